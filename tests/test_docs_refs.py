"""The documents cite measurement files under profiles/ as evidence: every file they name exists, and profiles/INDEX.md
lists every file that is there (regenerate with `python tools/profiles_index.py`)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = re.compile(r"(r0\d_[a-z0-9]+_[A-Za-z0-9_]+\.(?:jsonl|json|txt|log))")
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "HISTORY.md", "audio_amd/functional.py", "audio_amd/transforms.py",
        "audio_amd/csrc/melspec400.h", "audio_amd/csrc/lfilter_wave.h", "audio_amd/csrc/resample_mfma.h",
        "audio_amd/csrc/fftconv_fdr.h", "tools/bench_configs.py", "bench.py"]


def test_every_cited_profile_file_exists():
    have = set(os.listdir(os.path.join(ROOT, "profiles")))
    missing = {}
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        if not os.path.exists(path):
            continue
        with open(path, encoding="utf-8") as f:
            for name in PAT.findall(f.read()):
                if name not in have:
                    missing.setdefault(doc, set()).add(name)
    assert not missing, {d: sorted(v) for d, v in missing.items()}


def test_profiles_index_lists_every_file():
    with open(os.path.join(ROOT, "profiles", "INDEX.md"), encoding="utf-8") as f:
        index = f.read()
    absent = [n for n in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if n != "INDEX.md" and f"`{n}`" not in index]
    assert not absent, absent
