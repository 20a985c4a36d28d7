"""The C-ABI library loads without a GPU, exports every symbol include/sygnals_hip.h declares, and
rejects bad arguments before touching the device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sygnals_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(syg_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from sygnals_amd import _lib
    decl = declared_functions()
    assert len(decl) >= 15
    assert sorted(_lib.SIGNATURES) == decl


def test_library_exports_every_declared_symbol():
    from sygnals_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    h = C.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(h, name), f"{name} is declared in the header but not exported"
    assert _lib.lib().syg_abi_version() == 1


def test_argument_errors_are_reported_without_device_work():
    from sygnals_amd import _lib
    h = _lib.lib()
    rc = h.syg_stft2048_mel_f32(None, 1, 48000, 48000, 512, 1, 94, None, None, None, None, 40, None, 48000.0, 0.85, 2.0,
                                0, None, None, None, None)
    assert rc == -1 and b"null pointer" in h.syg_last_error()
    buf = (C.c_float * 16)()
    p = C.cast(buf, C.c_void_p)
    rc = h.syg_fft_pow2_c2c_f32(p, p, 1, 12, 0, p, None)
    assert rc == -1 and b"power of two" in h.syg_last_error()
    rc = h.syg_stft2048_c2c_f32(p, 1, 48000, 48000, 512, 1, 93, p, p, p, None)
    assert rc == -1 and b"framing rule" in h.syg_last_error()
    rc = h.syg_logmel_dct_f32(p, 1, 40, 94, p, 41, None, 1e-10, 80.0, 1, 1.0, None, p, None)
    assert rc == -1 and b"K <= M" in h.syg_last_error()
    sos = (C.c_double * 6)(1, 0, 0, 1, 0, 0); zi = (C.c_double * 2)(0, 0)
    rc = h.syg_sosfiltfilt_f32(p, 1, 20, 20, C.cast(sos, C.c_void_p), C.cast(zi, C.c_void_p), 1, 27, p, 20, p, None)
    assert rc == -1 and b"greater than padlen, which is 27" in h.syg_last_error()
    assert h.syg_sosfiltfilt_work_bytes(1024, 48000, 27, 4) == 0       # clip-resident form: no workspace
    assert h.syg_sosfiltfilt_work_bytes(4, 200000, 27, 4) > 0 and h.syg_sosfiltfilt_work_bytes(4, 48000, 27, 6) > 0
    assert h.syg_stft_mfcc_pow2_fits(1024, 40, 188, 13) == 1 and h.syg_stft_mfcc_pow2_fits(1024, 128, 1000, 13) == 0
    assert h.syg_sosfiltfilt_work_bytes(1, 100, 9, 9) == -1
    # partial sums per stream: clamp(8192 / B rounded down to a multiple of 4, 16, 2048)
    # float32 partial rows + the float64 slice sums of the two-stage combine (16 slices)
    assert h.syg_welch_work_bytes(8, 4096) == 8 * 1024 * 2049 * 4 + 8 * 16 * 2049 * 8
    assert h.syg_welch_work_bytes(1, 4096) == 1 * 2048 * 2049 * 4 + 1 * 16 * 2049 * 8
    assert h.syg_welch_work_bytes(1024, 256) == 1024 * 16 * 129 * 4 + 1024 * 16 * 129 * 8
    # the one-launch MFCC form: the library owns the LDS-fit rule
    assert h.syg_stft2048_mfcc_fits(40, 94, 13) == 2 and h.syg_stft2048_mfcc_fits(128, 313, 13) == 0
    assert h.syg_stft2048_mfcc_fits(128, 94, 13) == 1          # in the stage buffer's place
    assert h.syg_stft2048_mfcc_fits(40, 94, 41) == 0 and h.syg_stft2048_mfcc_fits(0, 94, 1) == 0
    with pytest.raises(_lib.SygnalsHipError, match="padlen"):
        _lib.check(-1, "x")


def test_header_is_valid_c_and_cpp(tmp_path):
    """include/sygnals_hip.h is what a maintainer binds against: it must compile as plain C99 and as C++ on its own,
    and a C translation unit that calls every entry point through it must link against the library."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = os.path.join(root, "include", "sygnals_hip.h")
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", "c", hdr], check=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", hdr], check=True)
    names = sorted(set(re.findall(r"\b(syg_[a-z0-9_]+)\s*\(", open(hdr).read())))
    src = tmp_path / "link.c"
    src.write_text('#include "sygnals_hip.h"\n#include <stdio.h>\nint main(void) {\n  const void* p[] = {'
                   + ", ".join(f"(const void*){n}" for n in names) + '};\n  printf("%d %d\\n", (int)(sizeof(p) / sizeof(p[0])), '
                   'syg_abi_version());\n  return 0;\n}\n')
    lib = os.path.join(root, "sygnals_amd", "lib")
    exe = tmp_path / "link"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-L", lib, "-lsygnals_hip",
                    f"-Wl,-rpath,{lib}", "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == len(names) and int(out[1]) == 1
