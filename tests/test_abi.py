"""CPU: the C-ABI library loads and exports every symbol include/nrm_hotpath.h declares; the host-side
argument validation works without a GPU (no compute call is made here)."""
import os
import re

from conftest import ROOT
from news_recommendation_model_amd import native


def header_functions():
    src = open(os.path.join(ROOT, "include", "nrm_hotpath.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nrm_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    names = header_functions()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in nrm_hotpath.h but not exported"
        assert n in native.SIGNATURES, f"{n} has no ctypes signature in native.py"
    assert sorted(native.SIGNATURES) == names


def test_abi_version_and_sizes(lib):
    assert lib.nrm_abi_version() == native.ABI_VERSION
    for D in (64, 72, 256, 400, 768):
        n16 = (D + 15) // 16
        assert lib.nrm_pwattn_packed_floats(D) >= n16 * 16 * n16 * 16
    assert lib.nrm_pwattn_bwd_nsplit(1024, 30, 50, 400, 0) >= 1
    assert lib.nrm_pwattn_bwd_nsplit(1, 1, 1, 64, 0) == 1


def test_in_tree_library_is_a_product_build(lib):
    """nrm_build_flags() is the OR of every timing-diagnostic override (scripts/_diag): 0 for the library the tests run."""
    assert lib.nrm_build_flags() == 0


def test_resident_w_backward_plan(lib):
    # bf16 arithmetics with D <= 256 have the resident-W backward; fp32 and wider attentions keep the E-form
    assert lib.nrm_pwattn_bwd_rw_supported(256, 2) == 1 and lib.nrm_pwattn_bwd_rw_supported(64, 1) == 1
    assert lib.nrm_pwattn_bwd_rw_supported(256, 0) == 0 and lib.nrm_pwattn_bwd_rw_supported(400, 2) == 0
    assert lib.nrm_pwattn_bwd_rw_packed_floats(256, 2) >= 256 * 256          # hi + lo bf16 images = 4 bytes per weight
    rc = lib.nrm_pwattn_bwd_contract(None, None, None, None, 256, None, None, None, 2, 3, 4, 64, 1, 2, 1, None)
    assert rc != 0 and b"NRM_DZ_HL4" in lib.nrm_last_error()                  # only the dW_p-only pass (4) reads NRM_DZ_HL4


def test_dp_walk_backward_plan(lib):
    # fp32 dP walk (round 5): histories of at least 16 rows, widths that are 16-byte multiples; the packed W_p^T image covers whole N-chunks
    assert lib.nrm_pwattn_bwd_dp_supported(400, 50) == 1 and lib.nrm_pwattn_bwd_dp_supported(64, 200) == 1
    assert lib.nrm_pwattn_bwd_dp_supported(400, 15) == 0 and lib.nrm_pwattn_bwd_dp_supported(66, 50) == 0
    assert lib.nrm_pwattn_bwd_dp_packed_floats(400, 50) == 25 * (2 * 13 * 16) * 16       # 25 K-chunks x two N-chunks of 13 tiles
    assert lib.nrm_pwattn_bwd_dp_packed_floats(768, 128) == 48 * (4 * 12 * 16) * 16      # four N-chunks of 12 tiles
    assert lib.nrm_pwattn_bwd_dp_packed_floats(64, 200) == 4 * (1 * 4 * 16) * 16
    assert lib.nrm_pwattn_bwd_dp_packed_floats(400, 8) == 0
    rc = lib.nrm_pwattn_bwd_dp_dtdh(None, None, None, None, None, None, 2, 3, 50, 400, None)
    assert rc != 0 and b"null" in lib.nrm_last_error()
    rc = lib.nrm_pwattn_bwd_dp_dtdh(None, None, None, None, None, None, 2, 3, 8, 400, None)
    assert rc != 0
    rc = lib.nrm_pwattn_bwd_dp_pack(None, 1600, 400, 50, None, None)
    assert rc != 0 and b"null" in lib.nrm_last_error()


def test_host_validation_rejects_bad_shapes(lib):
    # null pointers / bad D are refused on the host before any launch
    rc = lib.nrm_pwattn_fwd(None, None, None, None, None, None, None, None, None, 2, 3, 4, 66, 0, None)
    assert rc != 0 and b"multiple of 4" in lib.nrm_last_error()
    rc = lib.nrm_pwattn_fwd(None, None, None, None, None, None, None, None, None, 2, 3, 4, 64, 0, None)
    assert rc != 0 and b"null" in lib.nrm_last_error()
    rc = lib.nrm_pwattn_bwd_contract(None, None, None, None, 256, None, None, None, 1 << 20, 64, 64, 64, 3, 0, 0, None)
    assert rc != 0 and b"2^31" in lib.nrm_last_error()
    rc = lib.nrm_pwattn_fwd(None, None, None, None, None, None, None, None, None, 2, 3, 4, 64, 7, None)
    assert rc != 0 and b"mma" in lib.nrm_last_error()          # unknown arithmetic selector


def test_torch_library_ops_are_defined_without_a_gpu():
    """The nrm:: ops register at import time (schemas + fake functions need no device); on a CPU tensor they refuse."""
    import pytest
    import torch
    from news_recommendation_model_amd import ops
    for name in ops.OPS:
        assert hasattr(torch.ops.nrm, name)
    with pytest.raises(RuntimeError):
        ops.linear(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises((RuntimeError, NotImplementedError)):
        torch.ops.nrm.weighted_pool_fwd(torch.zeros(1, 2, 3), torch.zeros(1, 3, 4))


def test_gemm_tn_plans_of_the_c3_step_fill_one_round_of_wave_slots(lib):
    """Host-side plan of nrm_gemm_tn (no GPU needed): for the weight-gradient shapes of a C3 step the wave tile is the one with the
    fewest padded 16x16 tiles (2x8 / 8x2 for 402 x 1608, 5x5 for 400 x 400, 5x2 for 400 x 402 -- csrc/gemm.hip) and tiles x splits is
    one round of wave tasks at that kernel's occupancy: never more than the slots (a second, mostly empty round), never under 90 %."""
    cases = [  # ni, nj, R, wave tiles (of the chosen shape), waves per SIMD of that kernel
        (402, 1608, 30720, 13 * 13, 4), (1608, 402, 30720, 13 * 13, 4), (400, 400, 51200, 5 * 5, 3), (400, 400, 30720, 5 * 5, 3),
        (400, 402, 51200, 5 * 13, 4), (402, 400, 51200, 13 * 5, 4)]
    for ni, nj, R, tiles, wps in cases:
        ns = lib.nrm_gemm_tn_nsplit(ni, nj, R, 0)
        assert ns % 4 == 0 and 0.9 * 1024 * wps <= tiles * ns <= 1024 * wps, (ni, nj, R, ns)
    assert lib.nrm_gemm_tn_nsplit(64, 64, 3840, 0) == 30                  # small shapes: at least 128 rows per split
    assert lib.nrm_gemm_tn_nsplit(0, 64, 128, 0) == 0


def test_library_says_which_sources_it_was_built_from(lib):
    """Provenance: nrm_source_digest() of the loaded binary equals the digest of the kernel sources in the tree
    (what profiles/ are stamped with and bench.py prints), and native.load refuses a library built from other sources."""
    from news_recommendation_model_amd import build, native
    prov = native.provenance()
    assert prov["library_sources_sha256"] == build.sources_digest() and len(prov["library_sources_sha256"]) == 64
    assert "gfx950" in prov["build_info"] and "built 20" in prov["build_info"]
    assert build.library_digest() == build.sources_digest() and not build.needs_build()


def test_loader_refuses_a_library_built_from_other_sources(lib, monkeypatch):
    """native.load compares the binary's nrm_source_digest() with the kernel sources lying next to it: a tree whose csrc/ was
    edited without a rebuild (here: the digest function is made to report other sources) is refused with a message that says how
    to rebuild; NRM_ALLOW_STALE_LIB=1 and an explicit NRM_HOTPATH_LIB (variant builds of scripts/_diag) switch the check off."""
    import pytest
    from news_recommendation_model_amd import build, native
    monkeypatch.setattr(native, "_lib", None)
    monkeypatch.setattr(build, "sources_digest", lambda: "0" * 64)
    monkeypatch.delenv("NRM_ALLOW_STALE_LIB", raising=False)
    monkeypatch.delenv("NRM_HOTPATH_LIB", raising=False)
    with pytest.raises(RuntimeError, match="built from other kernel sources"):
        native.load()
    monkeypatch.setenv("NRM_ALLOW_STALE_LIB", "1")
    assert native.load().nrm_abi_version() == native.ABI_VERSION
