"""Column sharding for many independent bitmaps (BASELINE.json configs[4]).

Bitmap columns are unrelated, so a node with N GPUs is N independent compressors: column c belongs to rank
c mod N, every rank walks its own columns on its own device and streams, and the only cross-rank step is
adding up byte counts and taking the slowest rank's time.  No collective touches bitmap data.

The reference has no multi-device code at all (single device, default stream: compress.cu:129,166); this
module is the "many columns" driver a caller of the reference would have written around compress().
"""
from collections import namedtuple

ColumnSpec = namedtuple("ColumnSpec", "index kind seed n_words")

KINDS = ("sparse", "clustered", "dense")  # uniform p=0.01 / runs of mean 4096 bits / uniform p=0.5


def shard_columns(n_columns, rank, world):
    """Indices of the columns rank `rank` of `world` owns: c with c % world == rank (round robin)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return list(range(rank, n_columns, world))


def column_spec(index, n_words, seed=1337):
    """Deterministic description of synthetic column `index`: the three bench distributions in turn."""
    return ColumnSpec(index, KINDS[index % len(KINDS)], seed + index, n_words)


def make_column(wah, spec, device, out=None):
    """Generate the column in HBM on `device` (bit-exact definition: include/wah_gen.h); `out`: into this tensor."""
    if spec.kind == "sparse":
        return wah.gen_uniform_device(spec.n_words, spec.seed, 0.01, device=device, out=out)
    if spec.kind == "dense":
        return wah.gen_uniform_device(spec.n_words, spec.seed, 0.5, device=device, out=out)
    return wah.gen_clustered_device(spec.n_words, spec.seed, 4096, device=device, out=out)


SEGMENT_WORDS = 992  # one reference block: fills never cross it


def make_column_matrix(wah, specs, device):
    """The columns of `specs` (equal lengths, whole 992-word segments) as ONE tensor [len(specs), n_words]."""
    import torch

    n = specs[0].n_words
    if any(sp.n_words != n for sp in specs) or n % SEGMENT_WORDS:
        raise ValueError("a column matrix needs equal column lengths that are multiples of 992 words")
    m = torch.empty((len(specs), n), dtype=torch.int32, device=device)
    for row, sp in zip(m, specs):
        make_column(wah, sp, device, out=row)
    return m


def compress_column_matrix(compressor, matrix, wait=True):
    """All columns of a contiguous [columns, n_words] matrix in ONE launch (n_words a multiple of 992).

    Fills are maximal inside a 992-word segment and never cross one (F4), and every column is a whole number of
    segments, so compressing the matrix as one long bitmap yields exactly the columns' streams back to back; the
    segment index of the `indexed` compressor says where each one starts.  Returns (stream, column_offsets): column c
    is stream[column_offsets[c] : column_offsets[c + 1]], bit-identical to compressing it alone.  wait=False: only
    enqueue the launch (read compressor.result() / .seg_offsets after synchronising)."""
    columns, n = matrix.shape
    if n % SEGMENT_WORDS or not matrix.is_contiguous():
        raise ValueError("columns must be contiguous and a multiple of 992 words long")
    if compressor.seg_offsets is None:
        raise ValueError("needs DeviceCompressor(..., indexed=True)")
    compressor.run(matrix.view(-1))
    if not wait:
        return None, None
    stream = compressor.result()
    segs = n // SEGMENT_WORDS
    return stream, compressor.seg_offsets[:: segs][: columns + 1]


def compress_column_ranges(compressor, flat, lengths, wait=True):
    """Columns of DIFFERENT lengths (each a multiple of 992 words) stored back to back in `flat`: still one launch.
    Returns (stream, column_offsets) like compress_column_matrix: column c is stream[column_offsets[c] :
    column_offsets[c + 1]], bit-identical to compressing it alone (F4: a fill never crosses a 992-word segment)."""
    import torch

    if any(n % SEGMENT_WORDS for n in lengths) or sum(lengths) != flat.numel() or not flat.is_contiguous():
        raise ValueError("column lengths must be multiples of 992 words and add up to the buffer")
    if compressor.seg_offsets is None:
        raise ValueError("needs DeviceCompressor(..., indexed=True)")
    compressor.run(flat)
    if not wait:
        return None, None
    stream = compressor.result()
    first_segment = [0]
    for n in lengths:
        first_segment.append(first_segment[-1] + n // SEGMENT_WORDS)
    return stream, compressor.seg_offsets[torch.tensor(first_segment, device=compressor.seg_offsets.device)]


def compress_shards_multi_device(wah, shards):
    """Column shards over several GPUs by ONE call of the C ABI (include/wah.h: wah_compress_columns_multi_device): `shards` =
    list of (matrix, compressor) with matrix a contiguous [columns, n_words] tensor on the shard's device and compressor a
    DeviceCompressor(matrix.numel(), device=that device, indexed=True).  One host thread per shard inside the library, each on
    its own device and stream; nothing is exchanged.  Returns [(stream, column_offsets)] like compress_column_matrix."""
    import ctypes

    class Shard(ctypes.Structure):
        _fields_ = [("device", ctypes.c_int), ("n_columns", ctypes.c_uint64), ("d_in", ctypes.c_void_p), ("d_out", ctypes.c_void_p),
                    ("out_capacity_words", ctypes.c_uint64), ("d_out_words", ctypes.c_void_p), ("d_segment_offsets", ctypes.c_void_p),
                    ("d_workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t)]

    if not shards:
        return []
    n = shards[0][0].shape[1]
    arr = (Shard * len(shards))()
    for a, (matrix, comp) in zip(arr, shards):
        if matrix.shape[1] != n or n % SEGMENT_WORDS or not matrix.is_contiguous() or comp.seg_offsets is None:
            raise ValueError("shards need contiguous [columns, n_words] matrices of one column length (a multiple of 992) and indexed compressors")
        a.device = matrix.device.index or 0
        a.n_columns = matrix.shape[0]
        a.d_in, a.d_out, a.out_capacity_words = matrix.data_ptr(), comp.out.data_ptr(), comp.capacity
        a.d_out_words, a.d_segment_offsets = comp.count.data_ptr(), comp.seg_offsets.data_ptr()
        a.d_workspace, a.workspace_bytes = comp.workspace.data_ptr(), comp.ws_bytes
    import torch

    for matrix, _ in shards:  # (the call's streams are its own: what other streams still write into the inputs must be complete)
        torch.cuda.synchronize(matrix.device)
    status = (ctypes.c_int * len(shards))()
    rc = wah.lib().wah_compress_columns_multi_device(len(shards), ctypes.cast(arr, ctypes.c_void_p), n, ctypes.cast(status, ctypes.c_void_p))
    if rc != 0:
        raise wah.WahError(f"wah_compress_columns_multi_device: {rc} (per shard: {list(status)})")
    segs = n // SEGMENT_WORDS
    return [(comp.out[: int(comp.count.item())], comp.seg_offsets[:: segs][: matrix.shape[0] + 1]) for matrix, comp in shards]


def compress_columns(compressor, columns):
    """Enqueue one compress pass per column on the current stream; returns the list of compressed sizes
    (device tensors, read them after synchronising).  The compressor's output buffer is reused, so callers
    that need the words copy them out between columns."""
    sizes = []
    for col in columns:
        compressor.run(col)
        sizes.append(compressor.count.clone())
    return sizes


def aggregate_throughput(bytes_per_rank, seconds_per_rank):
    """Whole-job rate: all bytes / the slowest rank's time (what bench.py reports for N > 1)."""
    return sum(bytes_per_rank) / max(seconds_per_rank)
