"""scan and expansion of tools/decode_island_time.py's stream timed apart (wah_decompress_scan_device / _expand_device)"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
n = 268435200
d = torch.zeros(n, dtype=torch.int32, device="cuda")
for k in range(4):
    lo = 992 * (30000 + 60000 * k)
    d[lo: lo + 992 * 1000] = wah.gen_uniform_device(992 * 1000, 7 + k, 0.5)
comp = wah.DeviceCompressor(n, unsegmented=True)
comp.run(d)
st = comp.result().clone()
L = wah.lib()
wsb = int(L.wah_decompress_workspace_bytes(st.numel(), n + 1))
ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
out = torch.empty(n + 1, dtype=torch.int32, device="cuda")
info = torch.zeros(2, dtype=torch.int64, device="cuda")
def timed(f, reps=5):
    f(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
scan = lambda: L.wah_decompress_scan_device(st.data_ptr(), st.numel(), info.data_ptr(), ws.data_ptr(), wsb, None)
exp = lambda: L.wah_decompress_expand_device(st.data_ptr(), st.numel(), out.data_ptr(), n + 1, info.data_ptr(), ws.data_ptr(), wsb, None)
print("scan", round(timed(scan), 4), "ms")
scan()
print("expand", round(timed(exp), 4), "ms", "status", L.wah_decompress_status(ws.data_ptr(), None), "ok", bool(torch.equal(out[:n], d)))
seq = ws[4 * 194: 4 * 195].view(torch.int32).item()
print("list counters", ws[4 * 192: 4 * 195].view(torch.int32).tolist(), "flags with bit 1:", int((ws[:].view(torch.uint8) == 2).sum().item()))
