"""ctypes loader of tools/bin/libt4r_tools.so (tools/t4r_tools.hip: measurement infrastructure, not product code)."""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "bin", "libt4r_tools.so")
_LIB = None


def available():
    return os.path.exists(PATH)


def lib():
    global _LIB
    if _LIB is None:
        if not available():
            raise FileNotFoundError(f"{PATH} missing: run `python -m transformers4rec_amd.build` (or __graft_entry__.build())")
        L = ctypes.CDLL(PATH)
        P, I, Lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
        L.t4r_tools_copy.restype = I
        L.t4r_tools_copy.argtypes = [P, P, P, Lg, I, I]
        L.t4r_tools_occupy.restype = I
        L.t4r_tools_occupy.argtypes = [P, I, I, I, P, Lg, P]
        _LIB = L
    return _LIB


def _stream(stream=None):
    return ctypes.c_void_p((stream or torch.cuda.current_stream()).cuda_stream)


def copy(dst, src, mode=0, blocks=0, stream=None):
    """dst <- src with the float4 copy kernel (mode 1: non-temporal loads / stores)"""
    assert dst.numel() * dst.element_size() == src.numel() * src.element_size()
    rc = lib().t4r_tools_copy(_stream(stream), dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size(), mode, blocks)
    if rc != 0:
        raise RuntimeError(f"t4r_tools_copy failed ({rc})")


class Occupier:
    """k resident do-nothing workgroups on a side stream, from start() until stop() (or max_us): what an RCCL ring kernel does
    to the chip while its collective runs.  threads / lds_bytes shape the footprint per CU."""

    def __init__(self, k, threads=256, lds_bytes=16 * 1024, max_us=20000, device=None):
        self.k, self.threads, self.lds_bytes, self.max_us = int(k), threads, lds_bytes, max_us
        dev = device or torch.device("cuda", torch.cuda.current_device())
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seen = torch.zeros(1, dtype=torch.int32, device=dev)
        self.side = torch.cuda.Stream(device=dev)

    def start(self):
        """clears the flag on the current stream, then launches the workgroups on the side stream behind that point"""
        if self.k <= 0:
            return
        self.flag.zero_()
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        rc = lib().t4r_tools_occupy(_stream(self.side), self.k, self.threads, self.lds_bytes, self.flag.data_ptr(), self.max_us,
                                    self.seen.data_ptr())
        if rc != 0:
            raise RuntimeError(f"t4r_tools_occupy failed ({rc})")

    def stop(self):
        """sets the flag from the current stream (in stream order: the workgroups leave when the work enqueued so far is done)"""
        if self.k <= 0:
            return
        self.flag.fill_(1)

    def join(self):
        if self.k > 0:
            torch.cuda.current_stream().wait_stream(self.side)
