"""wah-mi355x: MI355X-native WAH bitmap compressor / decompressor.

Host-side mirror of the reference's operator interface for this path:

    compress(data)   <->  compress()   /root/reference/compress.h:12-18, compress.cu:41-209
    decompress(comp) <->  decompress() /root/reference/decompress.h:11-17, decompress.cu:18-141

Same argument meaning (host buffers of 32-bit words in, a new host buffer out,
sizes in words, three millisecond timings) and the same error behaviour (a
failed call returns nothing usable: here it raises WahError where the reference
returns NULL).  Everything is computed by the HIP library `libwah_hip.so`
through its C ABI (include/wah.h); there is NO CPU fallback -- importing works
without a GPU (so the ABI can be inspected), but every compute call fails
loudly if the library or the device is missing.

The directory name contains a hyphen, so import it with
    importlib.import_module("gpu-wah_amd")
or through the `gpu_wah_amd` shim module at the repository root.
"""
from . import columns  # noqa: F401
from . import report  # noqa: F401
from .api import (  # noqa: F401
    WahError,
    Timings,
    lib,
    lib_path,
    build,
    compress,
    decompress,
    host_cache_release,
    max_compressed_words,
    decoded_words,
    compress_device,
    decompress_device,
    decompress_segments_device,
    build_index_device,
    DeviceCompressor,
    DeviceDecompressor,
    validate_device,
    bitop_device,
    bitop_indexed_device,
    bitop_many_indexed_device,
    merge_fills_device,
    StreamReport,
    gen_uniform_device,
    gen_clustered_device,
    copy_device,
    threshold_for,
    version,
    ABI_SYMBOLS,
)
