"""Two builds of libwah_hip.so on one box, one process: the host decompress() (its device phase, as tools/report.py prints it) and
wah_decompress_device on uniform bitmaps of 992 MiB, one bit in 2^i.  Box-to-box differences of 4-5 % hide anything smaller when two
builds are measured in two gpurun calls; this does not have them.  Only entry points that both builds export are used.
usage: python tools/lib_ab.py OLD.so NEW.so [i ...]   (default 9..14)"""
import ctypes as C
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

wah = importlib.import_module("gpu-wah_amd")
n = 256 * 1024 * 992
u32p, u64p, f32p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_float)


def load(path):
    lib = C.CDLL(path)
    lib.wah_decompress.restype = C.c_void_p
    lib.wah_decompress.argtypes = [C.c_void_p, C.c_uint64, u64p, f32p, f32p, f32p]
    lib.wah_free.argtypes = [C.c_void_p]
    lib.wah_decompress_workspace_bytes.restype = C.c_size_t
    lib.wah_decompress_workspace_bytes.argtypes = [C.c_uint64, C.c_uint64]
    lib.wah_decompress_device.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.wah_decompress_status.argtypes = [C.c_void_p, C.c_void_p]
    return lib


libs = [(os.path.basename(p), load(p)) for p in sys.argv[1:3]]
for i in [int(x) for x in sys.argv[3:]] or list(range(9, 15)):
    d = wah.gen_uniform_device(n, 1337, 2.0 ** -i)
    comp = wah.DeviceCompressor(n)
    comp.run(d)
    stream = comp.result().clone()
    del comp
    host = np.ascontiguousarray(stream.cpu().numpy())
    c = int(stream.numel())
    row = []
    for name, lib in libs:
        # host entry: median device phase of 9 calls after 2 that fill the kept buffers
        t = []
        for k in range(11):
            ow, t1, t2, t3 = C.c_uint64(), C.c_float(), C.c_float(), C.c_float()
            p = lib.wah_decompress(host.ctypes.data, c, C.byref(ow), C.byref(t1), C.byref(t2), C.byref(t3))
            assert p
            if k == 0:
                got = np.ctypeslib.as_array(C.cast(p, u32p), shape=(int(ow.value),))
                assert np.array_equal(got[:n], d.cpu().numpy().view(np.uint32)), "host decompress differs"
            lib.wah_free(p)
            if k >= 2:
                t.append(t2.value)
        # device entry
        cap = n + 1
        ws = torch.zeros(int(lib.wah_decompress_workspace_bytes(c, cap)), dtype=torch.uint8, device="cuda")
        out = torch.empty(cap, dtype=torch.int32, device="cuda")
        info = torch.zeros(4, dtype=torch.int64, device="cuda")
        sp = torch.cuda.current_stream().cuda_stream
        run = lambda: lib.wah_decompress_device(stream.data_ptr(), c, out.data_ptr(), cap, info.data_ptr(), ws.data_ptr(), ws.numel(), sp)
        for _ in range(3):
            assert run() == 0
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            run()
        ev[1].record()
        torch.cuda.synchronize()
        assert lib.wah_decompress_status(ws.data_ptr(), sp) == 0 and torch.equal(out[:n], d)
        row.append(f"{name}: host {sorted(t)[len(t) // 2]:.4f}  device {ev[0].elapsed_time(ev[1]) / 20:.4f}")
        del ws, out
    print(f"2^-{i} ({32 / 31 * n / c:.1f} groups per word)  " + "   ".join(row), flush=True)
    del d, stream
