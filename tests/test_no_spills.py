"""The built gfx950 code object must not spill in the throughput kernels.

Round 2 found two regressions by reading the ISA by hand: the narrow instantiations of the f16 resampler held 30 x 16 B per lane in
a 128-register budget (536 B of scratch per thread; default-quality Resample pairs ran 4-5 x slower) and the overlap-save kernels
spilled hoisted LDS addresses (every reload an s_waitcnt vmcnt(0) that waits for the prefetched inputs).  Neither changes a result,
so no parity test sees it; this test reads the kernel descriptors of libaudio_amd.so (llvm-objdump --offloading + llvm-readelf
--notes) and fails when a kernel outside the allow list uses scratch.  No GPU needed."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
# kernels that are allowed scratch, with the most they may use (bytes per thread): general-order IIR scans of order >= 8 (rare
# shapes; orders 3 .. 8 run as second-order sections), the 2048-point Kaldi kernel, the one-tile biquad kernel's table builder
ALLOW = [(r"aamd14lfilter_kernelILi(8|12|16)E", 4200), (r"aamd2p217kaldi_pow2_kernelILi32E", 128),
         (r"aamd3lfw19lfilter_wave_kernel", 96), (r"aamd3lfw24lfilter_wave_pipe_kernel", 32),
         (r"aamd3lfw25lfilter_wave_mover_kernel", 32), (r"aamd4m40015istft400_kernel", 16),
         # (round 6: the tools-only instantiations -- AAMD_LFW_LAB, AAMD_RSM_LAB, AAMD_MFCC_LAB -- are compiled into
         # libaudio_amd_lab.so only; test_the_product_library_holds_no_lab_instantiation below)
         # round 5, 24 577 .. 32 768 taps: three delayed spectra in registers; one item-loop constant (a 64-bit bound) is parked in
         # scratch OUTSIDE the block-step loop (test_real_block_delay_line_steps_do_not_touch_scratch covers the loop)
         (r"aamd3fdr17delay_line_kernelILi4E", 16)]


def _kernels():
    from audio_amd import _build
    so = _build.OUT
    if not os.path.exists(so):
        pytest.skip("libaudio_amd.so is not built")
    if not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))):
        pytest.skip("llvm-objdump / llvm-readelf not in this image")
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "lib.so")
        shutil.copy(so, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=d)
        objs = [f for f in os.listdir(d) if "gfx950" in f]
        assert len(objs) == 1, os.listdir(d)
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(d, objs[0])], check=True,
                               capture_output=True, text=True).stdout
    out, name = {}, None
    for line in notes.splitlines():
        m = re.match(r"\s*\.name:\s+(\S+)", line)
        if m:
            name = m.group(1)
            out[name] = {}
            continue
        m = re.match(r"\s*\.(private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count|vgpr_count):\s+(\d+)", line)
        if m and name:
            out[name][m.group(1)] = int(m.group(2))
    return out


def test_throughput_kernels_do_not_spill():
    ks = _kernels()
    assert len(ks) > 100, len(ks)                      # the library's kernels were found at all
    bad = []
    for name, k in ks.items():
        scratch = k.get("private_segment_fixed_size", 0)
        if scratch == 0 and k.get("vgpr_spill_count", 0) == 0:
            continue
        limit = max([lim for pat, lim in ALLOW if re.search(pat, name)] or [0])
        if scratch > limit:
            bad.append((name, scratch, k.get("vgpr_spill_count")))
    assert not bad, bad


@pytest.mark.parametrize("pattern", [r"aamd4m40017melspec400_kernel", r"aamd3rsm19resample_f16_kernelILi\d+ELi0E",
                                     r"aamd3fco19overlap_save_kernel", r"aamd3fco23overlap_save_fdl_kernel", r"aamd3fdr17delay_line_kernelILi[123]E",
                                     r"aamd3lfw25lfilter_wave_mover_kernelILi0E", r"aamd2p216stft_pow2_kernel"])
def test_headline_kernels_have_no_scratch_at_all(pattern):
    """The kernels behind the BASELINE configs: zero bytes of scratch, zero spilled registers."""
    ks = {n: k for n, k in _kernels().items() if re.search(pattern, n)}
    assert ks, pattern
    for n, k in ks.items():
        assert k.get("private_segment_fixed_size", 0) == 0 and k.get("vgpr_spill_count", 0) == 0, (n, k)


def test_real_block_delay_line_steps_do_not_touch_scratch():
    """cfg5b's kernel (fdr::delay_line_kernel<NP>): no scratch instruction anywhere between the first and the last workgroup
    barrier of the disassembly -- the block-step loop; a reload there would wait for the prefetched inputs (s_waitcnt
    vmcnt(0)).  (The first build of the 3-partition instantiation sat at exactly 128 registers and parked item-loop constants
    in scratch; complete twiddle tables in LDS instead of formed powers brought it to 104.)"""
    from audio_amd import _build
    so = _build.OUT
    if not os.path.exists(so) or not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("library or llvm-objdump missing")
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "lib.so")
        shutil.copy(so, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True, cwd=d)
        obj = [f for f in os.listdir(d) if "gfx950" in f][0]
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(d, obj)], check=True,
                             capture_output=True, text=True).stdout
    for np_ in (2, 3, 4):
        m = re.search(r"<_ZN4aamd3fdr17delay_line_kernelILi%dE[^>]*>:\n(.*?)s_endpgm" % np_, dis, re.S)
        assert m, np_
        body = m.group(1).splitlines()
        bars = [i for i, l in enumerate(body) if "s_barrier" in l]
        # round 5: a step has 4 workgroup barriers (first pass, before / after the middle step, before the last pass) and two
        # rendezvous of wave pairs (s_sleep polls) where rounds 3-4 had 7 barriers; + the barrier behind the table set-up
        assert len(bars) == 5, len(bars)
        assert sum("s_sleep" in l for l in body) >= 2
        # the first barrier follows the table set-up; the step loop spans from the second barrier to the end of the last
        # pass: the stores that follow the last barrier belong to it
        inside = [l for l in body[bars[1]:] if "scratch_" in l]
        assert not inside, (np_, inside[:3])


def test_the_product_library_holds_no_lab_instantiation():
    """the tools-only kernel variants (first template argument != 0 of the biquad kernels, variants 1 / 2
    of the f16 resampler, the MFCC epilogue's lab instantiation) live in libaudio_amd_lab.so (-DAAMD_LAB), not in the product."""
    ks = _kernels()
    lab = [k for k in ks if re.search(r"aamd3lfw19lfilter_wave_kernelILi[1-9]", k)
           or re.search(r"aamd3lfw2[45]lfilter_wave_(pipe|mover)_kernelILi[1-9]", k)
           or re.search(r"aamd3rsm19resample_f16_kernelILi\d+ELi[12]E", k)
           or re.search(r"aamd4m40017melspec400_kernelILi524288E", k)]
    assert not lab, lab


def test_build_staleness_is_decided_by_content_not_by_modification_time():
    """touching a source must not rebuild, changing the flags must."""
    from audio_amd import _build
    if not os.path.exists(_build.OUT) or _build.stale():
        pytest.skip("libaudio_amd.so is not built from the current sources")
    src = os.path.join(_build.CSRC, "hd.h")
    st = os.stat(src)
    try:
        os.utime(src, None)                         # a newer mtime, the same bytes
        assert not _build.stale()
    finally:
        os.utime(src, (st.st_atime, st.st_mtime))
    keep = os.environ.get("AAMD_EXTRA_HIPCC_FLAGS")
    try:
        os.environ["AAMD_EXTRA_HIPCC_FLAGS"] = "-DSOMETHING_ELSE=1"
        assert _build.stale()
    finally:
        if keep is None:
            del os.environ["AAMD_EXTRA_HIPCC_FLAGS"]
        else:
            os.environ["AAMD_EXTRA_HIPCC_FLAGS"] = keep
