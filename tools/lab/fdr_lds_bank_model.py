import collections
pad=lambda i: i + ((i >> 5) << 1)
def rd(addrs):  # b64 read: two groups of 32 lanes, idx mod 32
    return sum(max(collections.Counter(a % 32 for a in set(g)).values()) for g in (addrs[:32], addrs[32:]))
def wr(addrs):  # b64 write: four groups of 16 lanes, idx mod 16
    return sum(max(collections.Counter(a % 16 for a in set(addrs[16*k:16*k+16])).values()) for k in range(4))
def m16(t):
    w, l = t >> 6, t & 63
    blk = 4*w + [0,2,1,3][l >> 4]; j = l & 15
    return blk, j
def m2(t):
    w, l = t >> 6, t & 63
    j = l & 1; i = (l >> 1) & 7; odd = (l >> 4) & 1; half = l >> 5
    return 32*w + 16*half + 2*i + odd, j
for wave in (0, 7, 15):
    tids = list(range(64*wave, 64*wave+64))
    p3 = [[pad(128*m16(t)[0] + m16(t)[1] + 16*r) for t in tids] for r in range(8)]
    p4 = [[pad(16*m2(t)[0] + m2(t)[1] + 2*r) for t in tids] for r in range(8)]
    p1 = [[pad(t + 1024*r) for t in tids] for r in range(8)]
    p2 = [[pad(1024*(t>>7) + (t&127) + 128*r) for t in tids] for r in range(8)]
    print(wave, "p1 rd/wr", sum(map(rd,p1)), sum(map(wr,p1)), "p2", sum(map(rd,p2)), sum(map(wr,p2)), "p3", sum(map(rd,p3)), sum(map(wr,p3)), "p4", sum(map(rd,p4)), sum(map(wr,p4)), "(ideal rd 16, wr 32)")
# coverage checks
assert sorted(m16(t) for t in range(1024)) == sorted((b, j) for b in range(64) for j in range(16))
assert sorted(m2(t) for t in range(1024)) == sorted((b, j) for b in range(512) for j in range(2))
assert all(m2(t)[0] == m2(t ^ 1)[0] and m2(t)[1] != m2(t^1)[1] for t in range(1024))
# wave-locality: pass 3 and 4 elements of a wave stay inside [512 w, 512 w + 512)
for t in range(1024):
    w = t >> 6
    assert 512*w <= 128*m16(t)[0] < 512*(w+1) and 512*w <= 16*m2(t)[0] < 512*(w+1)
print("ok")
import collections, itertools
pad=lambda i: i + ((i >> 5) << 1)
def rev_pos(k): return ((k & 7) << 10) + (((k >> 3) & 7) << 7) + (((k >> 6) & 7) << 4) + (((k >> 9) & 7) << 1) + (k >> 12)
def k_of_q(q): return (q >> 9) + 8*((q >> 6) & 7) + 64*((q >> 3) & 7) + 512*(q & 7)
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[x+32 for x in g] for g in G128]
def rd128(addrs):   # addrs: complex idx (even), 16-B slot = idx/2 mod 16
    return sum(max(collections.Counter((addrs[l]//2) % 16 for l in g).values()) for g in G128)
def wr64(addrs):
    return sum(max(collections.Counter(a % 16 for a in set(addrs[16*k:16*k+16])).values()) for k in range(4))
def cost(perm, waves=(0,3,9,15)):
    tot = 0
    for w in waves:
        a0s = {0: [], 1: []}; a1s = {0: [], 1: []}
        for l in range(64):
            bits = [(l >> b) & 1 for b in range(6)]
            v = sum(bits[perm[b]] << b for b in range(6))      # v: 6 bits -> h = bit0, c_local = bits 1..5
            h, cl = v & 1, v >> 1
            c = 32*w + cl
            for s in (0,1):
                q = 8*c + 2*h + s
                k = k_of_q(q)
                a0s[s].append(pad(2*q))
                a1s[s].append(pad(rev_pos(4096 - k)) if k else pad(8))
        for s in (0,1):
            tot += rd128(a0s[s]) + rd128(a1s[s])
            tot += wr64(a0s[s]) + wr64([a+1 for a in a0s[s]]) + wr64(a1s[s]) + wr64([a+1 for a in a1s[s]])
    return tot
ident = tuple(range(6))
print("identity", cost(ident), "ideal", 4*(2*(4+4) + 2*4*4))
best = sorted((cost(p), p) for p in itertools.permutations(range(6)))
print(best[:5])
def cost2(perm, cmap, waves=(0,3,9,15)):
    tot = 0
    for w in waves:
        a0s = {0: [], 1: []}; a1s = {0: [], 1: []}
        for l in range(64):
            bits = [(l >> b) & 1 for b in range(6)]
            v = sum(bits[perm[b]] << b for b in range(6))
            h, cl = v & 1, v >> 1
            c = cmap(w, cl)
            for s in (0,1):
                q = 8*c + 2*h + s
                k = k_of_q(q)
                a0s[s].append(pad(2*q))
                a1s[s].append(pad(rev_pos(4096 - k)) if k else pad(8))
        for s in (0,1):
            tot += rd128(a0s[s]) + rd128(a1s[s])
            tot += wr64(a0s[s]) + wr64([a+1 for a in a0s[s]]) + wr64(a1s[s]) + wr64([a+1 for a in a1s[s]])
    return tot
for name, cmap in (("16cl+w", lambda w, cl: 16*cl + w), ("w-low3", lambda w, cl: ((cl >> 2) << 6) | ((w) << 2) | (cl & 3)),
                   ("mix", lambda w, cl: ((cl & 7) << 6) | (w << 2) | (cl >> 3))):
    best = sorted((cost2(p, cmap), p) for p in itertools.permutations(range(6)))
    print(name, best[:3])
