#!/usr/bin/env python
"""Compile the REFERENCE's own native lfilter core (src/libtorchaudio/lfilter.cpp + utils.cpp, from where they
lie under /root/reference -- nothing is copied) into oracle/_ref/, and bind it the way the reference does
(`torchaudio.functional.filtering._lfilter_core_loop`, filtering.py:934-937).  TEST INFRASTRUCTURE: used by
tests/golden/make_grad_golden.py to generate gradient fixtures with the reference's real CPU kernel (the
pure-Python fallback loop updates its input in place and cannot be differentiated).  Only runs where
/root/reference exists (the build container); the GPU box uses the committed fixtures."""
import os
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def load():
    if not os.path.isdir(REF):
        raise RuntimeError("reference checkout not present")
    import torch
    from torch.utils import cpp_extension
    os.makedirs(OUT, exist_ok=True)
    src = [os.path.join(REF, "src/libtorchaudio", f) for f in ("lfilter.cpp", "utils.cpp")]
    cpp_extension.load(name="ref_libtorchaudio_lfilter", sources=src, extra_include_paths=[os.path.join(REF, "src")],
                       extra_cflags=["-O2", "-DTORCH_TARGET_VERSION=0x020a000000000000"], build_directory=OUT,
                       is_python_module=False, verbose=False)
    sys.path.insert(0, os.path.join(REF, "src"))
    import torchaudio.functional.filtering as filtering
    filtering._lfilter_core_loop = torch.ops.torchaudio._lfilter_core_loop
    return filtering


if __name__ == "__main__":
    load()
    print("oracle/_ref: reference lfilter core built and bound")
