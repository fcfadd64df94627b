"""The oracle under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; GPU sanitizers are not available on the
pool): the checker itself must not read or write out of bounds on the inputs of the parity matrix."""
import os
import subprocess
import sys

from _common import ROOT

DRIVER = r'''
import ctypes, sys, os
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import _common
O = ctypes.CDLL(os.path.join(%(root)r, "oracle", "libzada_oracle_asan.so"))
_common._cache["o"] = None
O.zo_deflate.argtypes = [ctypes.c_char_p, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64),
                         ctypes.POINTER(ctypes.c_uint32), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
n_ok = 0
for name, d in _common.edge_inputs().items():
    if len(d) > 400000 and name not in ("text_rand_text",):
        continue
    for m in (6, 7, 8, 10):
        out = ctypes.create_string_buffer(len(d) + 64)
        ol = ctypes.c_uint64(0); crc = ctypes.c_uint32(0xFFFFFFFF)
        rc = O.zo_deflate(d, len(d), m, out, len(d) + 64, ctypes.byref(ol), ctypes.byref(crc), None, None, None, None)
        assert rc in (0, 1), (name, m, rc)
        n_ok += 1
print("asan ok", n_ok)
'''


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libzada_oracle_asan.so"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
