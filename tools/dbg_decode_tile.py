"""Where the one-pass decoder's output differs from the bitmap: first mismatching words, their segments, how many segments are wrong.
usage: WAH_DT_BATCH=2 python tools/dbg_decode_tile.py [sparse|dense] [size_MiB]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
wah = importlib.import_module("gpu-wah_amd")
kind = sys.argv[1] if len(sys.argv) > 1 else "sparse"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = mib * 1024 * 1024 // 4 // 992 * 992
d = wah.gen_uniform_device(n, 1337, 0.01 if kind == "sparse" else 0.5)
comp = wah.DeviceCompressor(n, indexed=True)
comp.run(d)
stream = comp.result().clone()
offs = comp.seg_offsets[: n // 992 + 1].clone()
dec = wah.DeviceDecompressor(stream.numel(), n + 1)
for rep in range(3):
    dec.run(stream)
    back = dec.result()
    print("status", dec.status(), "decoded words", back.numel())
    m = min(back.numel(), n)
    bad = (back[:m] != d[:m]).nonzero().flatten()
    print(f"rep {rep}: mismatching words {bad.numel()}")
    if bad.numel():
        segs = torch.unique(bad // 992)
        print("  wrong segments:", segs.numel(), "first", segs[:12].tolist())
        tiles = torch.unique(offs[segs] // 8192)
        print("  tiles their first words lie in:", tiles.numel(), "min", int(tiles.min()), "max", int(tiles.max()), "list", tiles[:40].tolist())
        for sg in segs[:6].tolist():
            w = bad[(bad // 992) == sg]
            o0, o1 = int(offs[sg]), int(offs[sg + 1])
            print(f"  segment {sg}: {w.numel()} wrong words, first at word {int(w[0]) - sg * 992} of the segment; its stream words [{o0}, {o1}) = tile {o0 // 8192} + {o0 % 8192} .. tile {o1 // 8192} + {o1 % 8192}")
            i = int(w[0])
            print("    got ", [hex(x & 0xFFFFFFFF) for x in back[i : i + 4].tolist()], "want", [hex(x & 0xFFFFFFFF) for x in d[i : i + 4].tolist()])

# is a wrong segment the right data at the wrong place?  compare group streams of the first wrong segment at shifts -2000 .. 2000
import numpy as np
if bad.numel():
    sg = int(segs[0])
    lo = max(0, (sg - 3) * 992)
    want = d[lo : (sg + 4) * 992].cpu().numpy().view(np.uint32)
    got = back[sg * 992 : (sg + 1) * 992].cpu().numpy().view(np.uint32)
    wb = np.unpackbits(want.view(np.uint8), bitorder="little")
    gb = np.unpackbits(got.view(np.uint8), bitorder="little")
    off0 = (sg * 992 - lo) * 32
    best = None
    for sh in range(-2000 * 31, 2000 * 31 + 1, 31):
        a0 = off0 + sh
        if a0 < 0 or a0 + gb.size > wb.size:
            continue
        eq = int((wb[a0 : a0 + gb.size] == gb).sum())
        if best is None or eq > best[0]:
            best = (eq, sh // 31)
    print(f"segment {sg}: best alignment of its output with the bitmap: shift {best[1]} groups, {best[0]} of {gb.size} bits equal; ones in got {int(gb.sum())}, in want {int(wb[off0:off0+gb.size].sum())}")

# tile bases and granules in the workspace against what the stream says
P = int(os.environ.get("WAH_DT_BATCH", "2"))
W = dec.ws_bytes
half = ((W - 1024) // 2) & ~255
ws = dec.workspace
tb = ws[1024 + half : 1024 + half + 8 * ((stream.numel() + 4095) // 4096 + 1)].view(torch.int64)
w64 = stream.to(torch.int64) & 0xFFFFFFFF
grp = torch.where(w64 >= 0x80000000, w64 & 0x3FFFFFFF, torch.ones_like(w64))
cs = torch.cat([torch.zeros(1, dtype=torch.int64, device=grp.device), torch.cumsum(grp, 0)])
n_et = (stream.numel() + 4095) // 4096
exp = cs[torch.arange(0, n_et, device=grp.device) * 4096]
diff = (tb[:n_et] - exp)
wrong = diff.nonzero().flatten()
print("expand tiles with a wrong base:", wrong.numel(), "first", wrong[:16].tolist(), "differences", diff[wrong[:16]].tolist())
gran = ws[1024 : 1024 + 8 * 256 * 4].view(torch.int64)
nb = (stream.numel() + 8192 * P - 1) // (8192 * P)
tot = cs[torch.clamp(torch.arange(1, nb + 1, device=grp.device) * 8192 * P, max=stream.numel())] - cs[torch.arange(0, nb, device=grp.device) * 8192 * P]
k = min(nb, 1024)
gv = gran[:k] & ((1 << 48) - 1)
gw = (gv != tot[:k]).nonzero().flatten()
print("granules (first 1024 batches) that differ from the batch totals:", gw.numel(), gw[:16].tolist(), "epochs", torch.unique(gran[:k] >> 48).tolist())

# the row scan's view: row totals, slots in the workspace, and what the wrong bases are made of
rows = [int(tot[r * 256 : (r + 1) * 256].sum()) for r in range((nb + 255) // 256)]
slots = ws[1024 + 131072 * 1 : 1024 + 131072 + 8 * 8].view(torch.int64) if False else ws[1024 + 2 * 64 * 256 * 4 : 1024 + 2 * 64 * 256 * 4 + 64].view(torch.int64)
print("row totals", rows[:6])
print("slots (value, epoch)", [(int(x) & ((1 << 48) - 1), int(x) >> 48) for x in slots.tolist()])
batch_base = tb[: n_et : 2 * P]
for b in torch.unique(wrong // (2 * P))[:12].tolist():
    r, i = b // 256, b % 256
    exp_b = int(cs[b * 8192 * P])
    got_b = int(batch_base[b])
    sum_a = int(tot[r * 256 : b].sum())
    print(f"batch {b} (row {r}, idx {i}): expected {exp_b} = rows before {sum(rows[:r])} + own row part {sum_a}; got {got_b}; got - own row part = {got_b - sum_a}")
t1 = tot[256:512]
print("row 1 subsets: first 128", int(t1[:128].sum()), "last 128", int(t1[128:].sum()), "granules 0,1 of every lane", int(t1.view(64, 4)[:, :2].sum()), "granules 2,3", int(t1.view(64, 4)[:, 2:].sum()),
      "row 0 first 128", int(tot[:128].sum()), "row 0 last 128", int(tot[128:256].sum()))
c = torch.cat([torch.zeros(1, dtype=torch.int64, device=tot.device), torch.cumsum(tot, 0)]).cpu().numpy()
target = 2259851
import numpy as np
dm = c[None, :] - c[:, None]
hit = np.argwhere(dm == target)
print("contiguous batch ranges whose totals add up to", target, ":", hit[:10].tolist())
