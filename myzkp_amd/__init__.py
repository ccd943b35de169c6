"""myzkp_amd -- MI355X-native MSM / NTT prover path for MyZKP (Python binding of the C ABI, include/mzk.h).

The product is the HIP library myzkp_amd/libmzk_hip.so; this module only marshals numpy limb arrays
into it (ctypes).  There is no CPU fallback: importing works without a GPU (so the ABI can be
inspected), but every compute call raises MzkError(MZK_E_NOGPU) unless a gfx950 device is present.
"""
from ._lib import (MzkError, lib, FIELD_FR, FIELD_M128, FIELD_FQ, LIMBS, MODULUS, init, shutdown, set_workspace_budget, trim_workspace, workspace_bytes, init_devices, ctx_count, ctx_select, shard_range, msm_g1_multi, SrsMulti, ntt_multi, ntt_multi_dev, LAYOUT_CONTIGUOUS, LAYOUT_CYCLIC,
                   ntt, intt, ntt_batch, coset_lde, coset_lde_batch, poly_scale, fft_multiply, fast_multiply, root_of_unity, msm_g1,
                   kzg_setup_g1, kzg_commit, kzg_open, kzg_batch_open, kzg_prove_degree_bound, fri_fold, Srs, MerkleTree, merkle_commit_field, merkle_commit_field_batch, merkle_open_multi, fri_commit, fast_coset_divide, fast_zerofier, fast_evaluate, fast_interpolate, fast_interpolate_batch, fast_interpolate_batch_dev, msm_g2, kzg_setup_g2, g2_points_to_array, array_to_g2_points, to_limbs, from_limbs, points_to_array,
                   array_to_points, exported_symbols, DECLARED_SYMBOLS)

__all__ = ["MzkError", "lib", "FIELD_FR", "FIELD_M128", "FIELD_FQ", "LIMBS", "MODULUS", "init", "shutdown", "init_devices", "ctx_count", "ctx_select", "shard_range", "msm_g1_multi", "SrsMulti", "ntt_multi", "ntt_multi_dev", "LAYOUT_CONTIGUOUS", "LAYOUT_CYCLIC",
           "ntt", "intt", "ntt_batch", "coset_lde", "coset_lde_batch", "poly_scale", "fft_multiply", "fast_multiply", "root_of_unity", "msm_g1",
           "kzg_setup_g1", "kzg_commit", "kzg_open", "kzg_batch_open", "kzg_prove_degree_bound", "fri_fold", "Srs", "MerkleTree", "merkle_commit_field", "merkle_commit_field_batch", "merkle_open_multi", "fri_commit", "fast_coset_divide", "fast_zerofier", "fast_evaluate", "fast_interpolate", "fast_interpolate_batch", "fast_interpolate_batch_dev", "msm_g2", "kzg_setup_g2", "g2_points_to_array", "array_to_g2_points", "to_limbs", "from_limbs", "points_to_array",
           "array_to_points", "exported_symbols", "DECLARED_SYMBOLS"]
