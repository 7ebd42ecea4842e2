"""The C-ABI library loads and exports every symbol include/omnihip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "omni-video_amd", "omnivideo", "libomnihip.so")
HDR = os.path.join(ROOT, "include", "omnihip.h")


def declared_symbols():
    src = open(HDR).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ovd_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "ovd_flash_attn_fwd_bf16" in syms and "ovd_gemm_bf16" in syms and len(syms) >= 14


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "build with `make -C omni-video_amd/csrc` or __graft_entry__.build()"
    lib = ctypes.CDLL(LIB)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in omnihip.h but not exported"
    lib.ovd_abi_version.restype = ctypes.c_int
    assert lib.ovd_abi_version() >= 1


def test_binding_table_matches_header():
    import omnivideo._hip as H
    bound = set(H.SIGNATURES) | {"ovd_last_error", "ovd_abi_version"}
    assert bound == set(declared_symbols())


def test_argument_validation_without_gpu():
    """Rejected calls return non-zero with a message and launch nothing."""
    lib = ctypes.CDLL(LIB)
    lib.ovd_last_error.restype = ctypes.c_char_p
    rc = lib.ovd_gemm_bf16(None, 0, None, 0, None, None, 0, None, 1, 128, 64, 0, None)
    assert rc != 0 and b"null" in lib.ovd_last_error()


def test_product_path_has_no_cpu_fallback():
    import torch
    import pytest
    import omnivideo._hip as H
    with pytest.raises(ValueError):
        H.layernorm_modulate(torch.zeros(4, 256), torch.zeros(4, 256), 1e-6)


def test_peer_exchange_entry_points_validate_without_gpu():
    """The stream-ordered entry points of the peer-mapped exchange reject bad groups, chunk sizes and
    address tables before any launch."""
    lib = ctypes.CDLL(LIB)
    lib.ovd_last_error.restype = ctypes.c_char_p
    p, i, l, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
    lib.ovd_peer_barrier.argtypes = [p, p, p, i, i, d, p]
    lib.ovd_peer_all_to_all.argtypes = [p, p, l, i, i, p]
    lib.ovd_peer_sum_f32.argtypes = [p, p, l, i, p]
    lib.ovd_gather_col_blocks_from_bf16.argtypes = [p, p, l, l, i, i, i, p]
    buf = (ctypes.c_ubyte * 256)()
    a = ctypes.addressof(buf)
    assert lib.ovd_peer_barrier(None, a, a, 2, 0, 1.0, None) != 0 and b"null" in lib.ovd_last_error()
    assert lib.ovd_peer_barrier(a, a, a, 0, 0, 1.0, None) != 0 and b"bad group" in lib.ovd_last_error()
    assert lib.ovd_peer_barrier(a, a, a, 2, 2, 1.0, None) != 0 and b"bad group" in lib.ovd_last_error()
    assert lib.ovd_peer_barrier(a, a, a, 2, 0, 0.0, None) != 0 and b"timeout" in lib.ovd_last_error()
    assert lib.ovd_peer_all_to_all(a, a, 24, 2, 0, None) != 0 and b"multiple of 16" in lib.ovd_last_error()
    assert lib.ovd_peer_all_to_all(a, a, 32, 65, 0, None) != 0 and b"bad group" in lib.ovd_last_error()
    assert lib.ovd_peer_sum_f32(a, a, -1, 2, None) != 0
    assert lib.ovd_gather_col_blocks_from_bf16(a, a, 8, 4, 2, 8, 0, None) != 0       # row stride < nblk * col_block
    assert lib.ovd_gather_col_blocks_from_bf16(a, a, 16, 4, 2, 8, 2, None) != 0      # first block out of range


def test_group_communicator_kind_is_validated():
    import pytest
    from omnivideo.modules.engine import make_group_comm
    with pytest.raises(ValueError, match="auto, rccl, host or peer"):
        make_group_comm(None, "shm")
