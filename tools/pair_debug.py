"""small structured inputs through the device compressor against the oracle, first mismatches printed"""
import importlib, sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
wah = importlib.import_module("gpu-wah_amd")
from tests import _oracle
o = _oracle.load()
def check(name, data):
    data = np.ascontiguousarray(data, dtype=np.uint32)
    want = o.compress(data)
    d = torch.from_numpy(data.view(np.int32)).cuda()
    got = wah.compress_device(d).cpu().numpy().view(np.uint32)
    ok = np.array_equal(got, want)
    print(name, "OK" if ok else "MISMATCH", len(got), len(want))
    if not ok:
        n = min(len(got), len(want))
        bad = np.nonzero(got[:n] != want[:n])[0][:8]
        for i in bad: print("   word", i, "got %08x want %08x" % (got[i], want[i]))
rng = np.random.default_rng(5)
check("zeros 992", np.zeros(992))
check("zeros 64", np.zeros(64))
check("zeros 31", np.zeros(31))
check("zeros 62", np.zeros(62))
check("zeros 1984", np.zeros(1984))
check("ones 992", np.full(992, 0xFFFFFFFF))
z = np.zeros(992, dtype=np.uint32); z[500] = 1
check("one bit", z)
z = np.zeros(1984, dtype=np.uint32); z[5] = 1; z[700] = 4; z[1500] = 8
check("three bits", z)
check("random sparse 5000", (rng.random(5000) < 0.05).astype(np.uint32) * rng.integers(1, 2**32, 5000, dtype=np.uint64).astype(np.uint32))
check("random dense 3000", rng.integers(0, 2**32, 3000, dtype=np.uint64).astype(np.uint32))
