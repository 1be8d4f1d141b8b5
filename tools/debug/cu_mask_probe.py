"""How does a CU-masked HIP stream (hipExtStreamCreateWithCUMask, co_occ_amd.streams) behave on this MI355X?  One Winograd split-f16 layer
(128 -> 128, 80 000 rows: three kernels that fill the chip) timed on a plain stream and on streams masked to CUs [lo, hi)."""
import ctypes, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from co_occ_amd import core, _lib

dev = torch.device("cuda:0")
lib = _lib.load()
n = ctypes.c_int(0)
_lib.check(lib.coocc_device_cu_count(ctypes.byref(n)))
ncu = n.value
print("CUs:", ncu)


def masked(bits):
    nwords = (ncu + 31) // 32
    words = [0] * nwords
    for cu in bits:
        words[cu // 32] |= 1 << (cu % 32)
    arr = (ctypes.c_uint32 * nwords)(*words)
    out = ctypes.c_void_p(0)
    _lib.check(lib.coocc_stream_create_cu_mask(arr, nwords, ctypes.byref(out)))
    return torch.cuda.ExternalStream(out.value, device=dev)


g = torch.Generator().manual_seed(1)
x = core.to_rows(torch.randn(1, 128, 100, 100, 8, generator=g).to(dev))
pc = core.PackedConv((torch.randn(128, 128, 3, 3, 3, generator=g) * 0.02).to(dev), ksize=3, pad=1)


def timeit(st, label):
    with torch.cuda.stream(st), torch.no_grad():
        for _ in range(3):
            core.conv_rows(x, pc, relu=False)
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            core.conv_rows(x, pc, relu=False)
        e1.record()
        st.synchronize()
    print("%-48s %.1f us per layer" % (label, e0.elapsed_time(e1) * 1e3 / 20), flush=True)


timeit(torch.cuda.Stream(device=dev), "plain stream")
timeit(masked(range(ncu)), "mask: all %d bits" % ncu)
timeit(masked(range(ncu - 4)), "mask: bits [0, %d)" % (ncu - 4))
timeit(masked(range(4, ncu)), "mask: bits [4, %d)" % ncu)
timeit(masked(range(ncu // 2)), "mask: bits [0, %d)" % (ncu // 2))
timeit(masked(range(0, ncu, 2)), "mask: even bits")
timeit(masked(range(32)), "mask: bits [0, 32)")
timeit(masked([i for i in range(ncu) if i % 8 != 7 or i < ncu - 32]), "mask: all but bits 8k+7 of the last 32")
