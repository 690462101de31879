"""How fast the box WRITES 1 GiB (no reads): torch fill kernels and hipMemsetAsync, against decode_expand_kernel's clustered case
(4.4 M stream words in, 1 GiB out)."""
import torch
n = 268435200
x = torch.empty(n, dtype=torch.int32, device="cuda")
for name, fn in (("x.zero_()", lambda: x.zero_()), ("x.fill_(-1)", lambda: x.fill_(-1)), ("x.copy_(y) (read + write)", None)):
    if fn is None:
        y = torch.ones_like(x)
        fn = lambda: x.copy_(y)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): fn()
    ev[1].record(); torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 20
    moved = 4 * n * (2 if "copy" in name else 1)
    print(f"{name:28s}: {ms:.4f} ms  {moved / ms / 1e6:.0f} GB/s")
