import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
wah = importlib.import_module("gpu-wah_amd")
rng = np.random.default_rng(5)
sizes = [992 * 40, 17, 992 * 300 + 5, 0, 992 * 40, 2_000_000, 31]
cases = [(rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32) & np.uint32(0x01010000) if i % 2 else
          rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)) for i, n in enumerate(sizes)]
for rnd in range(2):
    for i, a in enumerate(cases):
        comp = wah.compress(a)
        print("round", rnd, "case", i, "n", a.size, "C", comp.size, flush=True)
        back = wah.decompress(comp)
        print("   decoded", back.size, "ok", bool(np.array_equal(back[: a.size], a)), flush=True)
    wah.host_cache_release()
print("done")
