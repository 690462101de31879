"""ctypes binding of libwah_hip.so (C ABI: include/wah.h).

PyTorch is used only as plumbing for device memory and streams in the
device-pointer helpers; no torch type crosses the ABI (plain pointers, sizes
and a hipStream_t passed as void*).
"""
import collections
import ctypes
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# WAH_LIB_PATH selects another build of the same library (tools/ use it for the diagnostic build)
_LIB_PATH = os.environ.get("WAH_LIB_PATH") or os.path.join(_HERE, "libwah_hip.so")

Timings = collections.namedtuple("Timings", "to_device_ms device_ms from_device_ms")

_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_f32p = ctypes.POINTER(ctypes.c_float)
_vp = ctypes.c_void_p
_u64 = ctypes.c_uint64
_sz = ctypes.c_size_t
_int = ctypes.c_int

# name -> (restype, argtypes): every extern "C" symbol include/wah.h declares
ABI_SYMBOLS = {
    "wah_compress": (_vp, [_vp, _u64, _u64p, _f32p, _f32p, _f32p]),
    "wah_decompress": (_vp, [_vp, _u64, _u64p, _f32p, _f32p, _f32p]),
    "wah_free": (None, [_vp]),
    "wah_host_cache_release": (None, []),
    "wah_max_compressed_words": (_u64, [_u64]),
    "wah_decoded_words": (_u64, [_u64]),
    "wah_compress_workspace_bytes": (_sz, [_u64]),
    "wah_decompress_workspace_bytes": (_sz, [_u64, _u64]),
    "wah_workspace_init_device": (_int, [_vp, _sz, _vp]),
    "wah_compress_device": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_compress_device_ex": (_int, [_vp, _u64, _vp, _u64, _vp, ctypes.c_uint, _vp, _sz, _vp]),
    "wah_compress_device_indexed": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _vp, _sz, _vp]),
    "wah_compress_status": (_int, [_vp, _vp]),
    "wah_compress_columns_multi_device": (_int, [_int, _vp, _u64, _vp]),
    "wah_decompress_device": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_decompress_device_ex": (_int, [_vp, _u64, _vp, _u64, _vp, ctypes.c_uint, _vp, _sz, _vp]),
    "wah_decompress_scan_device": (_int, [_vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_decompress_expand_device": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_build_index_device": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_decompress_segments_workspace_bytes": (_sz, []),
    "wah_decompress_segments_device": (ctypes.c_int, [_vp, _u64, _vp, _u64, _u64, _u64, _vp, _u64, _vp, _sz, _vp]),
    "wah_decompress_status": (_int, [_vp, _vp]),
    "wah_validate_device": (_int, [_vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_merge_fills_workspace_bytes": (_sz, [_u64]),
    "wah_merge_fills_device": (_int, [_vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_bitop_scratch_bytes": (_sz, [_u64, _u64, _u64]),
    "wah_bitop_device": (_int, [_int, _u64, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp, _sz, _vp]),
    "wah_bitop_status": (_int, [_vp, _u64, _u64, _u64, _vp]),
    "wah_bitop_indexed_scratch_bytes": (_sz, [_u64]),
    "wah_bitop_indexed_device": (_int, [_int, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _u64, _vp, _vp, _vp, _sz, _vp]),
    "wah_bitop_indexed_status": (_int, [_vp, _u64, _vp]),
    "wah_bitop_many_indexed_device": (_int, [_int, _u64, _int, _vp, _vp, _vp, _vp, _u64, _vp, _vp, _vp, _sz, _vp]),
    "wah_gen_uniform_device": (_int, [_vp, _u64, _u64, _u64, _vp]),
    "wah_gen_clustered_device": (_int, [_vp, _u64, _u64, _u64, _vp]),
    "wah_copy_device": (_int, [_vp, _vp, _u64, _vp]),
    "wah_last_decode_route": (_int, []),
    "wah_last_bitop_route": (_int, []),
    "wah_last_error": (ctypes.c_char_p, []),
    "wah_version": (ctypes.c_char_p, []),
}
# the reference's own C++-linkage symbols (compress.h:12-18, decompress.h:11-17)
CXX_SYMBOLS = ("_Z8compressPjyPyPfS1_S1_", "_Z10decompressPjyPyPfS1_S1_")


class WahError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def build(force=False, verbose=False):
    """Compile libwah_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc"))]
    srcs += [os.path.join(_HERE, "..", "include", f) for f in ("wah.h", "wah_gen.h", "compress.h", "decompress.h")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if stale:
        cmd = ["make", "-C", _HERE, "libwah_hip.so"] + (["-B"] if force else [])
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    """Load the HIP library; fail loudly when it is missing (there is no fallback path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise WahError(f"{_LIB_PATH} not found: build it with __graft_entry__.build() or "
                           f"`make -C {_HERE}` -- the HIP extension is required, there is no CPU fallback")
        # PyTorch-ROCm ships its own libamdhip64.so (same soname).  Import it first when it is installed, so
        # that this process has ONE HIP runtime: loading ours first makes torch map a second copy, and the
        # second runtime to initialise cannot create events or streams.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = ctypes.CDLL(_LIB_PATH)
        for name, (res, args) in ABI_SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def version():
    return lib().wah_version().decode()


def _err():
    return lib().wah_last_error().decode()


def _check(rc, what):
    if rc != 0:
        raise WahError(f"{what} failed (code {rc}): {_err()}")


def max_compressed_words(n_words):
    return int(lib().wah_max_compressed_words(int(n_words)))


def decoded_words(n_groups):
    return int(lib().wah_decoded_words(int(n_groups)))


def threshold_for(p):
    """Generator threshold for bit density p (include/wah_gen.h)."""
    return min(int(p * 2**32), 2**32)


# --------------------------------------------------------------------------
# host-pointer operators: the reference's API
# --------------------------------------------------------------------------
def _host_call(fn, data):
    a = np.ascontiguousarray(data, dtype=np.uint32)
    n_out = _u64(0)
    t = [ctypes.c_float(0.0) for _ in range(3)]
    ptr = fn(a.ctypes.data if a.size else None, a.size, ctypes.byref(n_out), ctypes.byref(t[0]), ctypes.byref(t[1]),
             ctypes.byref(t[2]))
    if not ptr:
        raise WahError(f"{fn.__name__} returned NULL: {_err()}")
    if n_out.value == 0:
        lib().wah_free(ptr)
        return np.empty(0, dtype=np.uint32), Timings(*(x.value for x in t))
    # hand the library's buffer out as it is (no second copy of up to a bitmap); freed when the array goes away
    buf = (ctypes.c_uint32 * n_out.value).from_address(ptr)
    weakref.finalize(buf, lib().wah_free, ptr)
    return np.frombuffer(buf, dtype=np.uint32), Timings(*(x.value for x in t))


def compress(data, with_timings=False):
    """compress() of the reference (compress.cu:41-209): host bitmap words in, compressed words out."""
    out, t = _host_call(lib().wah_compress, data)
    return (out, t) if with_timings else out


def decompress(comp, with_timings=False):
    """decompress() of the reference (decompress.cu:18-141): returns the ceil(31*G/32) decoded words."""
    out, t = _host_call(lib().wah_decompress, comp)
    return (out, t) if with_timings else out


# --------------------------------------------------------------------------
# device-pointer operators (torch tensors as device memory; int32 storage)
# --------------------------------------------------------------------------
def host_cache_release():
    """Return the device buffers compress()/decompress() keep between calls (include/wah.h: wah_host_cache_release)."""
    lib().wah_host_cache_release()


def _torch():
    import torch

    if not torch.cuda.is_available():
        raise WahError("no GPU visible: the device-pointer API needs an MI355X")
    return torch


def _stream_ptr(torch, stream=None):
    s = stream if stream is not None else torch.cuda.current_stream()
    return ctypes.c_void_p(s.cuda_stream)


def _as_words(torch, t):
    if t.dtype not in (torch.int32, torch.uint32) or not t.is_cuda or not t.is_contiguous():
        raise WahError("expected a contiguous 32-bit integer CUDA tensor")
    return t


class DeviceCompressor:
    """Reusable workspace + output for compressing bitmaps of up to `n_words` words in HBM.

    Replaces the per-call cudaMalloc/cudaFree of compress.cu:89-103,191-193: nothing is
    allocated inside run(), so it can be timed (and graph-captured) as pure device work.
    """

    def __init__(self, n_words, device="cuda:0", indexed=False, unsegmented=False, no_wait=False):
        """no_wait: the three-launch route in which no workgroup waits for another (include/wah.h: WAH_NO_WAIT)."""
        torch = _torch()
        if indexed and unsegmented:
            raise WahError("an unsegmented stream has no segment index")
        if no_wait and indexed:
            raise WahError("the flags entry point has no index output (WAH_FORCE_FALLBACK=1 covers the indexed call)")
        self.unsegmented = bool(unsegmented)
        self.no_wait = bool(no_wait)
        self.n_words = int(n_words)
        self.capacity = max_compressed_words(self.n_words)
        self.ws_bytes = int(lib().wah_compress_workspace_bytes(self.n_words))
        # zeroed once (include/wah.h: wah_workspace_init_device); the kernel keeps it up from then on
        self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=device)
        self.out = torch.empty(max(self.capacity, 1), dtype=torch.int32, device=device)
        self.count = torch.zeros(1, dtype=torch.int64, device=device)
        n_seg = (self.capacity + 1023) // 1024
        self.seg_offsets = torch.zeros(n_seg + 1, dtype=torch.int64, device=device) if indexed else None

    def run(self, d_in, n_words=None, stream=None, count=None):
        """Enqueue one compress pass; returns nothing (read .count / .out after synchronising).  count: another
        one-element int64 device tensor to receive C instead of .count (a slot per column, say)."""
        torch = _torch()
        _as_words(torch, d_in)
        n = self.n_words if n_words is None else int(n_words)
        if n > self.n_words or n > d_in.numel():
            raise WahError("input larger than this compressor was sized for")
        sp = _stream_ptr(torch, stream)
        if count is not None:
            if count.dtype != torch.int64 or count.numel() != 1 or not count.is_cuda:
                raise WahError("count must be a one-element int64 device tensor")
            return self._run(d_in, n, sp, count.data_ptr())
        return self._run(d_in, n, sp, self.count.data_ptr())

    def _run(self, d_in, n, sp, count_ptr):
        if self.unsegmented or self.no_wait:
            rc = lib().wah_compress_device_ex(d_in.data_ptr(), n, self.out.data_ptr(), self.capacity, count_ptr,
                                              (1 if self.unsegmented else 0) | (2 if self.no_wait else 0),
                                              self.workspace.data_ptr(), self.ws_bytes, sp)
        elif self.seg_offsets is None:
            rc = lib().wah_compress_device(d_in.data_ptr(), n, self.out.data_ptr(), self.capacity, count_ptr,
                                           self.workspace.data_ptr(), self.ws_bytes, sp)
        else:
            rc = lib().wah_compress_device_indexed(d_in.data_ptr(), n, self.out.data_ptr(), self.capacity,
                                                   count_ptr, self.seg_offsets.data_ptr(),
                                                   self.workspace.data_ptr(), self.ws_bytes, sp)
        _check(rc, "wah_compress_device")

    def status(self, stream=None):
        torch = _torch()
        _check(lib().wah_compress_status(self.workspace.data_ptr(), _stream_ptr(torch, stream)), "compress")

    def result(self, stream=None):
        """Synchronise, check the launch status, return the compressed words as a device tensor view."""
        self.status(stream)
        return self.out[: int(self.count.item())]


class DeviceDecompressor:
    """Reusable workspace + output for decoding streams of up to `c_words` words into `out_capacity` words."""

    ROUTES = {0: "none", 1: "one pass", 2: "two launches", 3: "no wait"}

    def __init__(self, c_words, out_capacity_words, device="cuda:0", no_wait=False, two_launches=False, one_pass=False):
        """no_wait: the sums pass by the route in which no workgroup waits for another (include/wah.h: WAH_NO_WAIT);
        two_launches / one_pass: that decoder whatever the capacity suggests (WAH_TWO_LAUNCHES / WAH_ONE_PASS).  After
        run(): `route` = the decoder the library launched (wah_last_decode_route)."""
        torch = _torch()
        self.no_wait = bool(no_wait)
        self.two_launches = bool(two_launches)
        self.one_pass = bool(one_pass)
        self.route = "none"
        self.c_words = int(c_words)
        self.capacity = int(out_capacity_words)
        self.ws_bytes = int(lib().wah_decompress_workspace_bytes(self.c_words, self.capacity))
        # zeroed once (include/wah.h: wah_workspace_init_device); the sums kernel keeps it up from then on
        self.workspace = torch.zeros(self.ws_bytes, dtype=torch.uint8, device=device)
        self.out = torch.empty(max(self.capacity, 1), dtype=torch.int32, device=device)
        self.info = torch.zeros(2, dtype=torch.int64, device=device)  # [decoded words, groups]

    def run(self, d_comp, c_words=None, stream=None):
        torch = _torch()
        _as_words(torch, d_comp)
        c = self.c_words if c_words is None else int(c_words)
        if c > self.c_words or c > d_comp.numel():
            raise WahError("stream larger than this decompressor was sized for")
        rc = lib().wah_decompress_device_ex(d_comp.data_ptr(), c, self.out.data_ptr(), self.capacity, self.info.data_ptr(),
                                            (2 if self.no_wait else 0) | (4 if self.two_launches else 0) | (8 if self.one_pass else 0),
                                            self.workspace.data_ptr(), self.ws_bytes, _stream_ptr(torch, stream))
        self.route = self.ROUTES.get(int(lib().wah_last_decode_route()), "?")
        _check(rc, "wah_decompress_device")

    def status(self, stream=None):
        torch = _torch()
        _check(lib().wah_decompress_status(self.workspace.data_ptr(), _stream_ptr(torch, stream)), "decompress")

    def result(self, stream=None):
        self.status(stream)
        return self.out[: int(self.info[0].item())]


def compress_device(d_in):
    """One-shot device compress: int32 CUDA tensor in, compressed int32 CUDA tensor out."""
    c = DeviceCompressor(d_in.numel(), device=d_in.device)
    c.run(d_in)
    return c.result().clone()


def decompress_device(d_comp, out_capacity_words):
    d = DeviceDecompressor(d_comp.numel(), out_capacity_words, device=d_comp.device)
    d.run(d_comp)
    return d.result().clone()


def build_index_device(d_comp):
    """Segment index (int64 tensor, ceil(G / 1024) + 1 entries) of a stream of compress() that came without one, and its
    group count G (wah_build_index_device).  Raises for streams that have no such index."""
    torch = _torch()
    _as_words(torch, d_comp)
    c = int(d_comp.numel())
    ws_bytes = int(lib().wah_decompress_workspace_bytes(c, 0))
    workspace = torch.zeros(ws_bytes, dtype=torch.uint8, device=d_comp.device)
    info = torch.zeros(2, dtype=torch.int64, device=d_comp.device)
    # a stream of c words has at most c segments (every segment holds at least one word)
    offsets = torch.zeros(c + 1, dtype=torch.int64, device=d_comp.device)
    sp = _stream_ptr(torch)
    _check(lib().wah_build_index_device(d_comp.data_ptr(), c, offsets.data_ptr(), offsets.numel(), info.data_ptr(),
                                        workspace.data_ptr(), ws_bytes, sp), "wah_build_index_device")
    _check(lib().wah_decompress_status(workspace.data_ptr(), sp), "build_index")
    groups = int(info[1].item())
    return offsets[: (groups + 1023) // 1024 + 1].clone(), groups


def decompress_segments_device(d_comp, seg_offsets, n_words, first_segment=0, n_segments=None, out=None, workspace=None,
                               check=True):
    """Segments [first_segment, first_segment + n_segments) of the bitmap (992 words each, the last one shorter) from a
    stream of an `indexed` compressor and its seg_offsets, without scanning the stream (wah_decompress_segments_device).
    `out` / `workspace`: reuse these tensors; check=False: only enqueue (the caller reads the status later)."""
    torch = _torch()
    _as_words(torch, d_comp)
    groups = max_compressed_words(int(n_words))
    all_segments = (groups + 1023) // 1024
    if n_segments is None:
        n_segments = all_segments - first_segment
    total = decoded_words(groups)
    need = max(0, min((first_segment + n_segments) * 992, total) - first_segment * 992) if n_segments else 0
    if out is None:
        out = torch.empty(max(need, 1), dtype=torch.int32, device=d_comp.device)
    ws_bytes = int(lib().wah_decompress_segments_workspace_bytes())
    if workspace is None:
        workspace = torch.empty(ws_bytes, dtype=torch.uint8, device=d_comp.device)
    rc = lib().wah_decompress_segments_device(d_comp.data_ptr(), d_comp.numel(), seg_offsets.data_ptr(), int(n_words),
                                              int(first_segment), int(n_segments), out.data_ptr(), out.numel(),
                                              workspace.data_ptr(), workspace.numel(), _stream_ptr(torch))
    _check(rc, "wah_decompress_segments_device")
    if check:
        _check(lib().wah_decompress_status(workspace.data_ptr(), _stream_ptr(torch)), "decompress_segments")
    return out[:need]


def merge_fills_device(d_comp):
    """The stream with adjacent fills of one kind merged and empty fills dropped: unsegmented WAH (wah_merge_fills_device)."""
    torch = _torch()
    _as_words(torch, d_comp)
    c = int(d_comp.numel())
    ws_bytes = int(lib().wah_merge_fills_workspace_bytes(c))
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=d_comp.device)
    out = torch.empty(max(c, 1), dtype=torch.int32, device=d_comp.device)
    count = torch.zeros(1, dtype=torch.int64, device=d_comp.device)
    sp = _stream_ptr(torch)
    _check(lib().wah_merge_fills_device(d_comp.data_ptr(), c, out.data_ptr(), c, count.data_ptr(), ws.data_ptr(), ws_bytes, sp),
           "wah_merge_fills_device")
    _check(lib().wah_decompress_status(ws.data_ptr(), sp), "merge_fills")
    return out[: int(count.item())].clone()


OPS = {"and": 0, "or": 1, "xor": 2, "andnot": 3}


def bitop_device(op, d_a, d_b, n_words):
    """compress(A op B) from the two compressed bitmaps of n_words words each (include/wah.h: wah_bitop_device)."""
    torch = _torch()
    _as_words(torch, d_a)
    _as_words(torch, d_b)
    n, ca, cb = int(n_words), int(d_a.numel()), int(d_b.numel())
    cap = max_compressed_words(n)
    sc_bytes = int(lib().wah_bitop_scratch_bytes(n, ca, cb))
    scratch = torch.empty(sc_bytes, dtype=torch.uint8, device=d_a.device)
    out = torch.empty(max(cap, 1), dtype=torch.int32, device=d_a.device)
    count = torch.zeros(1, dtype=torch.int64, device=d_a.device)
    sp = _stream_ptr(torch)
    _check(lib().wah_bitop_device(OPS[op], n, d_a.data_ptr(), ca, d_b.data_ptr(), cb, out.data_ptr(), cap, count.data_ptr(),
                                  scratch.data_ptr(), sc_bytes, sp), "wah_bitop_device")
    _check(lib().wah_bitop_status(scratch.data_ptr(), n, ca, cb, sp), "bitop")
    return out[: int(count.item())].clone()


def bitop_indexed_device(op, d_a, a_offsets, d_b, b_offsets, n_words, scratch=None, out=None, out_offsets=None, check=True):
    """compress(A op B) from two compressed bitmaps that come with their segment indexes (wah_bitop_indexed_device).
    Returns (stream, seg_offsets) of the result; scratch / out / out_offsets: reuse these tensors; check=False: only
    enqueue and return (out, count tensor, out_offsets)."""
    torch = _torch()
    _as_words(torch, d_a)
    _as_words(torch, d_b)
    n = int(n_words)
    cap = max_compressed_words(n)
    n_seg = (cap + 1023) // 1024
    sc_bytes = int(lib().wah_bitop_indexed_scratch_bytes(n))
    if scratch is None:
        scratch = torch.empty(sc_bytes, dtype=torch.uint8, device=d_a.device)
    if out is None:
        out = torch.empty(max(cap, 1), dtype=torch.int32, device=d_a.device)
    if out_offsets is None:
        out_offsets = torch.zeros(n_seg + 1, dtype=torch.int64, device=d_a.device)
    count = torch.zeros(1, dtype=torch.int64, device=d_a.device)
    sp = _stream_ptr(torch)
    _check(lib().wah_bitop_indexed_device(OPS[op], n, d_a.data_ptr(), d_a.numel(), a_offsets.data_ptr(), d_b.data_ptr(),
                                          d_b.numel(), b_offsets.data_ptr(), out.data_ptr(), out.numel(), count.data_ptr(),
                                          out_offsets.data_ptr(), scratch.data_ptr(), scratch.numel(), sp),
           "wah_bitop_indexed_device")
    if not check:
        return out, count, out_offsets
    _check(lib().wah_bitop_indexed_status(scratch.data_ptr(), n, sp), "bitop_indexed")
    return out[: int(count.item())], out_offsets


def bitop_many_indexed_device(op, operands, n_words, scratch=None, out=None, out_offsets=None, check=True):
    """compress(A op B op C ...) for up to 8 (stream, seg_offsets) pairs in one combining pass
    (wah_bitop_many_indexed_device).  Returns as bitop_indexed_device."""
    torch = _torch()
    k = len(operands)
    for st, _ in operands:
        _as_words(torch, st)
    dev = operands[0][0].device
    n = int(n_words)
    cap = max_compressed_words(n)
    n_seg = (cap + 1023) // 1024
    sc_bytes = int(lib().wah_bitop_indexed_scratch_bytes(n))
    if scratch is None:
        scratch = torch.empty(sc_bytes, dtype=torch.uint8, device=dev)
    if out is None:
        out = torch.empty(max(cap, 1), dtype=torch.int32, device=dev)
    if out_offsets is None:
        out_offsets = torch.zeros(n_seg + 1, dtype=torch.int64, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    streams = (ctypes.c_void_p * k)(*[st.data_ptr() for st, _ in operands])
    words = (ctypes.c_uint64 * k)(*[st.numel() for st, _ in operands])
    offsets = (ctypes.c_void_p * k)(*[o.data_ptr() for _, o in operands])
    sp = _stream_ptr(torch)
    _check(lib().wah_bitop_many_indexed_device(OPS[op], n, k, streams, words, offsets, out.data_ptr(), out.numel(),
                                               count.data_ptr(), out_offsets.data_ptr(), scratch.data_ptr(), scratch.numel(), sp),
           "wah_bitop_many_indexed_device")
    if not check:
        return out, count, out_offsets
    _check(lib().wah_bitop_indexed_status(scratch.data_ptr(), n, sp), "bitop_many_indexed")
    return out[: int(count.item())], out_offsets


StreamReport = collections.namedtuple(
    "StreamReport", "groups words empty_fills fillable_literals crossing_fills unmerged_fills segment_canonical")


def validate_device(d_comp):
    """What a compressed stream contains (include/wah.h: wah_validate_device), without decoding it."""
    torch = _torch()
    _as_words(torch, d_comp)
    c = int(d_comp.numel())
    ws_bytes = int(lib().wah_decompress_workspace_bytes(c, 0))
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=d_comp.device)
    report = torch.zeros(8, dtype=torch.int64, device=d_comp.device)
    _check(lib().wah_validate_device(d_comp.data_ptr(), c, report.data_ptr(), ws.data_ptr(), ws_bytes, _stream_ptr(torch)),
           "wah_validate_device")
    _check(lib().wah_decompress_status(ws.data_ptr(), _stream_ptr(torch)), "validate")
    r = report.cpu().tolist()
    return StreamReport(r[0], r[1], r[2], r[3], r[4], r[5], bool(r[6]))


def gen_uniform_device(n_words, seed, p, device="cuda:0", out=None):
    """Bernoulli(p) bitmap generated in HBM (bit-exact definition: include/wah_gen.h); `out`: write into this tensor."""
    torch = _torch()
    if out is None:
        out = torch.empty(max(int(n_words), 1), dtype=torch.int32, device=device)
    else:
        _as_words(torch, out)
    _check(lib().wah_gen_uniform_device(out.data_ptr(), int(n_words), int(seed), threshold_for(p),
                                        _stream_ptr(torch)), "wah_gen_uniform_device")
    return out[: int(n_words)]


def gen_clustered_device(n_words, seed, mean_run_bits=4096, device="cuda:0", out=None):
    torch = _torch()
    if out is None:
        out = torch.empty(max(int(n_words), 1), dtype=torch.int32, device=device)
    else:
        _as_words(torch, out)
    _check(lib().wah_gen_clustered_device(out.data_ptr(), int(n_words), int(seed), threshold_for(1.0 / mean_run_bits),
                                          _stream_ptr(torch)), "wah_gen_clustered_device")
    return out[: int(n_words)]


def copy_device(d_in, d_out):
    torch = _torch()
    _check(lib().wah_copy_device(d_in.data_ptr(), d_out.data_ptr(), int(d_in.numel()), _stream_ptr(torch)),
           "wah_copy_device")
