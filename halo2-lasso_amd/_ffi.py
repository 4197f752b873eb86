"""ctypes binding of liblasso_hip.so -- the declarations of include/lasso_hip.h.

Loading fails loudly when the library has not been built (`__graft_entry__.build()` or
`make -C halo2-lasso_amd/csrc`): there is no Python or CPU fallback for any compute call.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblasso_hip.so")

LH_OK = 0
LH_ERR_INVALID_SUMCHECK, LH_ERR_INVALID_PCS_PARAM, LH_ERR_INVALID_PCS_OPEN = -1, -2, -3
LH_ERR_INVALID_SNARK, LH_ERR_SERIALIZATION, LH_ERR_TRANSCRIPT, LH_ERR_DEVICE, LH_ERR_ARG = -4, -5, -6, -7, -8
LH_SC_EVALUATIONS, LH_SC_COEFFICIENTS = 0, 1
LH_SC_MAX_TERMS, LH_SC_MAX_FACTORS = 48, 4
LH_LASSO_MAX_CHUNKS, LH_LASSO_MAX_MEMORIES, LH_LASSO_MAX_TERMS = 8, 16, 16
LH_LASSO_NUM_PHASES = 9


class lh_fr(C.Structure):
    _fields_ = [("l", C.c_uint64 * 4)]


class lh_g1(C.Structure):
    _fields_ = [("x", C.c_uint64 * 4), ("y", C.c_uint64 * 4)]


class lh_sop(C.Structure):
    _fields_ = [("num_terms", C.c_uint32), ("global_eq", C.c_int32),
                ("coeff", lh_fr * LH_SC_MAX_TERMS),
                ("num_factors", C.c_uint8 * LH_SC_MAX_TERMS),
                ("factor", (C.c_uint8 * LH_SC_MAX_FACTORS) * LH_SC_MAX_TERMS)]


class lh_expr_node(C.Structure):
    _fields_ = [("op", C.c_uint32), ("a", C.c_int32), ("b", C.c_int32), ("reserved", C.c_uint32), ("scalar", lh_fr)]


class lh_expr(C.Structure):
    _fields_ = [("nodes", C.POINTER(lh_expr_node)), ("num_nodes", C.c_size_t)]


class lh_lasso_table(C.Structure):
    _fields_ = [("num_chunks", C.c_uint32), ("chunk_bits", C.c_uint32), ("num_memories", C.c_uint32),
                ("memory_chunk", C.c_uint32 * LH_LASSO_MAX_MEMORIES),
                ("memory_subtable", C.c_uint32 * LH_LASSO_MAX_MEMORIES),
                ("num_terms", C.c_uint32),
                ("g_coeff", lh_fr * LH_LASSO_MAX_TERMS),
                ("g_num_factors", C.c_uint8 * LH_LASSO_MAX_TERMS),
                ("g_factor", (C.c_uint8 * LH_SC_MAX_FACTORS) * LH_LASSO_MAX_TERMS)]


class lh_hp_lasso_lookup(C.Structure):
    _fields_ = [("table", lh_lasso_table), ("output_poly", C.c_size_t), ("chunk_polys", C.c_size_t * LH_LASSO_MAX_CHUNKS)]


class lh_hp_lookup(C.Structure):
    _fields_ = [("inputs", C.POINTER(lh_expr)), ("tables", C.POINTER(lh_expr)), ("width", C.c_size_t)]


class lh_hp_param(C.Structure):
    _fields_ = [("num_vars", C.c_size_t),
                ("num_instance_polys", C.c_size_t), ("num_instances", C.POINTER(C.c_size_t)),
                ("num_preprocess_polys", C.c_size_t), ("d_preprocess_polys", C.POINTER(C.c_void_p)),
                ("num_witness_polys", C.c_size_t), ("num_challenges", C.c_size_t),
                ("num_lookups", C.c_size_t), ("lookups", C.POINTER(lh_hp_lookup)),
                ("num_permutation_polys", C.c_size_t), ("permutation_poly_index", C.POINTER(C.c_size_t)),
                ("d_permutation_polys", C.POINTER(C.c_void_p)),
                ("num_permutation_z_polys", C.c_size_t),
                ("expression", lh_expr),
                ("num_lasso_lookups", C.c_size_t), ("lasso_lookups", C.POINTER(lh_hp_lasso_lookup))]


_SYNTH_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.POINTER(lh_fr), C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t)


class lh_hp_circuit(C.Structure):
    _fields_ = [("user", C.c_void_p), ("synthesize", _SYNTH_CB)]


class lh_hp_vparam(C.Structure):
    _fields_ = [("num_vars", C.c_size_t),
                ("num_instance_polys", C.c_size_t), ("num_instances", C.POINTER(C.c_size_t)),
                ("num_witness_polys", C.c_size_t), ("num_challenges", C.c_size_t),
                ("num_lookups", C.c_size_t), ("num_permutation_z_polys", C.c_size_t),
                ("expression", lh_expr),
                ("num_preprocess_polys", C.c_size_t), ("preprocess_comms", C.POINTER(lh_g1)),
                ("num_permutation_polys", C.c_size_t), ("permutation_comms", C.POINTER(lh_g1)),
                ("num_lasso_lookups", C.c_size_t), ("lasso_lookups", C.POINTER(lh_hp_lasso_lookup))]


class lh_evaluation(C.Structure):
    _fields_ = [("poly", C.c_uint32), ("point", C.c_uint32), ("value", lh_fr)]


_AG_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
_AGD_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
LH_RCCL_UNIQUE_ID_BYTES = 128


class lh_comm(C.Structure):
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("user", C.c_void_p), ("all_gather", _AG_CB),
                ("all_gather_device", _AGD_CB)]


class lh_prof_rec(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("ms", C.c_double), ("bytes", C.c_double), ("muls", C.c_double),
                ("items", C.c_double)]


_FE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(lh_fr))
_G1_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(lh_g1))


class lh_transcript(C.Structure):
    _fields_ = [("user", C.c_void_p),
                ("write_field_element", _FE_CB), ("common_field_element", _FE_CB),
                ("squeeze_challenge", _FE_CB),
                ("write_commitment", _G1_CB), ("common_commitment", _G1_CB),
                ("read_field_element", _FE_CB), ("read_commitment", _G1_CB)]


class lh_g2(C.Structure):
    _fields_ = [("x_c0", C.c_uint64 * 4), ("x_c1", C.c_uint64 * 4), ("y_c0", C.c_uint64 * 4), ("y_c1", C.c_uint64 * 4)]


_P = C.c_void_p
_SZ = C.c_size_t
# name -> (restype, argtypes); every symbol include/lasso_hip.h declares
SIGNATURES = {
    "lh_last_error": (C.c_char_p, []),
    "lh_version": (C.c_char_p, []),
    "lh_ctx_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "lh_ctx_destroy": (None, [_P]),
    "lh_ctx_sync": (C.c_int, [_P]),
    "lh_ctx_stream": (_P, [_P]),
    "lh_alloc": (C.c_int, [_P, _SZ, C.POINTER(_P)]),
    "lh_free": (C.c_int, [_P, _P]),
    "lh_upload": (C.c_int, [_P, _P, _P, _SZ]),
    "lh_download": (C.c_int, [_P, _P, _P, _SZ]),
    "lh_keccak_transcript_new": (C.c_int, [C.POINTER(C.POINTER(lh_transcript))]),
    "lh_keccak_transcript_free": (None, [C.POINTER(lh_transcript)]),
    "lh_keccak_transcript_proof": (C.c_int, [C.POINTER(lh_transcript), C.POINTER(C.POINTER(C.c_uint8)),
                                             C.POINTER(_SZ)]),
    "lh_fr_from_u64": (C.c_int, [_P, _P, _SZ, _P]),
    "lh_fr_from_u32": (C.c_int, [_P, _P, _SZ, _P]),
    "lh_fr_to_repr": (C.c_int, [_P, _P, _SZ, _P]),
    "lh_fr_from_repr": (C.c_int, [_P, _P, _SZ, _P]),
    "lh_fr_add": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "lh_fr_sub": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "lh_fr_mul": (C.c_int, [_P, _P, _P, _SZ, _P]),
    "lh_fr_mul_chain": (C.c_int, [_P, _P, _P, _SZ, C.c_int, _P]),
    "lh_fr_batch_invert": (C.c_int, [_P, _P, _SZ, _P]),
    "lh_fix_var": (C.c_int, [_P, _P, _SZ, C.POINTER(lh_fr), _P]),
    "lh_eq_xy": (C.c_int, [_P, C.POINTER(lh_fr), _SZ, _P]),
    "lh_evaluate": (C.c_int, [_P, C.POINTER(_P), _SZ, _SZ, C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_lincomb": (C.c_int, [_P, C.POINTER(_P), C.POINTER(lh_fr), _SZ, _SZ, _P]),
    "lh_sumcheck_prove": (C.c_int, [_P, C.c_int, _SZ, C.POINTER(lh_sop), C.POINTER(_P), _SZ,
                                    C.POINTER(lh_fr), _SZ, C.POINTER(lh_fr), C.POINTER(lh_transcript),
                                    C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_sumcheck_prove_expr": (C.c_int, [_P, _SZ, C.POINTER(lh_expr), C.POINTER(_P), _SZ, C.POINTER(lh_fr), _SZ,
                                         C.POINTER(lh_fr), _SZ, C.POINTER(lh_fr), C.POINTER(lh_transcript),
                                         C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_gkr_fractional_prove": (C.c_int, [_P, _SZ, _SZ, C.POINTER(C.POINTER(lh_fr)), C.POINTER(C.POINTER(lh_fr)),
                                          C.POINTER(_P), C.POINTER(_P), C.POINTER(lh_transcript),
                                          C.POINTER(lh_fr), C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_grand_product_prove": (C.c_int, [_P, _SZ, C.POINTER(_P), C.POINTER(_SZ), C.POINTER(lh_transcript),
                                         C.POINTER(lh_fr), C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_msm": (C.c_int, [_P, _P, _P, _SZ, C.POINTER(lh_g1)]),
    "lh_msm_u32": (C.c_int, [_P, _P, _P, _SZ, C.POINTER(lh_g1)]),
    "lh_mkzg_setup": (C.c_int, [_P, C.POINTER(lh_fr), _SZ, C.POINTER(_P)]),
    "lh_srs_upload": (C.c_int, [_P, _P, _SZ, C.POINTER(_P)]),
    "lh_srs_download": (C.c_int, [_P, _P, _P]),
    "lh_srs_num_vars": (_SZ, [_P]),
    "lh_srs_free": (None, [_P, _P]),
    "lh_mkzg_commit": (C.c_int, [_P, _P, _P, _SZ, C.POINTER(lh_g1)]),
    "lh_mkzg_batch_commit": (C.c_int, [_P, _P, C.POINTER(_P), _SZ, _SZ, C.POINTER(lh_g1)]),
    "lh_mkzg_open": (C.c_int, [_P, _P, _P, _SZ, C.POINTER(lh_fr), C.POINTER(lh_transcript), C.POINTER(lh_fr)]),
    "lh_mkzg_batch_open": (C.c_int, [_P, _P, _SZ, C.POINTER(_P), _SZ, C.POINTER(lh_fr), _SZ,
                                     C.POINTER(lh_evaluation), _SZ, C.POINTER(lh_transcript)]),
    "lh_lasso_prove": (C.c_int, [_P, _P, C.POINTER(lh_lasso_table), _SZ, C.POINTER(_P), C.POINTER(lh_transcript)]),
    "lh_lasso_last_timing": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "lh_ctx_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "lh_ctx_get_option": (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int64)]),
    "lh_lasso_last_route": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "lh_ctx_set_comm": (C.c_int, [_P, C.POINTER(lh_comm), _SZ]),
    "lh_rccl_unique_id": (C.c_int, [C.c_char_p]),
    "lh_ctx_set_comm_rccl": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, _SZ]),
    "lh_ctx_set_comm_loopback": (C.c_int, [_P, C.c_int, C.c_int, _SZ]),
    "lh_ctx_comm_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "lh_ctx_comm_phase_stats": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int]),
    "lh_ctx_memory_stats": (C.c_int, [_P, C.POINTER(C.c_uint64)]),
    "lh_ctx_host_cpus": (C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "lh_lasso_prove_sharded": (C.c_int, [_P, _P, C.POINTER(lh_lasso_table), _SZ, C.POINTER(_P),
                                         C.POINTER(lh_transcript)]),
    "lh_hyperplonk_prove": (C.c_int, [_P, _P, C.POINTER(lh_hp_param), C.POINTER(C.POINTER(lh_fr)), C.POINTER(_P),
                                      C.POINTER(lh_transcript)]),
    "lh_shard_extract": (C.c_int, [_P, _P, _SZ, _SZ, _SZ, _SZ, _SZ, _P]),
    "lh_hyperplonk_prove_sharded": (C.c_int, [_P, _P, C.POINTER(lh_hp_param), C.POINTER(C.POINTER(lh_fr)), C.POINTER(_P),
                                              C.POINTER(lh_transcript)]),
    "lh_hyperplonk_prove_phases": (C.c_int, [_P, _P, C.POINTER(lh_hp_param), _SZ, C.POINTER(_SZ), C.POINTER(_SZ),
                                             C.POINTER(C.POINTER(lh_fr)), C.POINTER(lh_hp_circuit),
                                             C.POINTER(lh_transcript)]),
    "lh_hyperplonk_verify_phases": (C.c_int, [_P, C.POINTER(lh_hp_vparam), _SZ, C.POINTER(_SZ), C.POINTER(_SZ),
                                              C.POINTER(C.POINTER(lh_fr)), C.POINTER(lh_transcript)]),
    "lh_keccak_transcript_from_proof": (C.c_int, [C.c_char_p, _SZ, C.POINTER(C.POINTER(lh_transcript))]),
    "lh_keccak_transcript_remaining": (C.c_int, [C.POINTER(lh_transcript), C.POINTER(_SZ)]),
    "lh_mkzg_vp_setup": (C.c_int, [C.POINTER(lh_fr), _SZ, C.POINTER(_P)]),
    "lh_mkzg_vp_new": (C.c_int, [C.POINTER(lh_g1), C.POINTER(lh_g2), C.POINTER(lh_g2), _SZ, C.POINTER(_P)]),
    "lh_mkzg_vp_export": (C.c_int, [_P, C.POINTER(lh_g1), C.POINTER(lh_g2), C.POINTER(lh_g2)]),
    "lh_mkzg_vp_num_vars": (_SZ, [_P]),
    "lh_mkzg_vp_free": (None, [_P]),
    "lh_pairing_check": (C.c_int, [C.POINTER(lh_g1), C.POINTER(lh_g2), _SZ, C.POINTER(C.c_int)]),
    "lh_mkzg_verify": (C.c_int, [_P, C.POINTER(lh_g1), C.POINTER(lh_fr), _SZ, C.POINTER(lh_fr),
                                 C.POINTER(lh_transcript)]),
    "lh_mkzg_batch_verify": (C.c_int, [_P, _SZ, C.POINTER(lh_g1), _SZ, C.POINTER(lh_fr), _SZ,
                                       C.POINTER(lh_evaluation), _SZ, C.POINTER(lh_transcript)]),
    "lh_sumcheck_verify": (C.c_int, [C.c_int, _SZ, _SZ, C.POINTER(lh_fr), C.POINTER(lh_transcript),
                                     C.POINTER(lh_fr), C.POINTER(lh_fr)]),
    "lh_lasso_verify": (C.c_int, [_P, C.POINTER(lh_lasso_table), _SZ, C.POINTER(lh_transcript)]),
    "lh_hyperplonk_verify": (C.c_int, [_P, C.POINTER(lh_hp_vparam), C.POINTER(C.POINTER(lh_fr)),
                                       C.POINTER(lh_transcript)]),
    "lh_ukzg_setup": (C.c_int, [_P, C.POINTER(lh_fr), _SZ, C.POINTER(_P)]),
    "lh_usrs_upload": (C.c_int, [_P, C.c_char_p, _SZ, C.POINTER(_P)]),
    "lh_usrs_download": (C.c_int, [_P, _P, C.c_char_p]),
    "lh_usrs_size": (_SZ, [_P]),
    "lh_usrs_free": (None, [_P, _P]),
    "lh_zeromorph_batch_commit": (C.c_int, [_P, _P, _SZ, C.POINTER(_P), _SZ, _SZ, C.POINTER(lh_g1)]),
    "lh_zeromorph_open": (C.c_int, [_P, _P, _SZ, _P, _SZ, C.POINTER(lh_fr), C.POINTER(lh_transcript)]),
    "lh_zeromorph_batch_open": (C.c_int, [_P, _P, _SZ, _SZ, C.POINTER(_P), _SZ, C.POINTER(lh_fr), _SZ,
                                          C.POINTER(lh_evaluation), _SZ, C.POINTER(lh_transcript)]),
    "lh_zeromorph_vp_setup": (C.c_int, [C.POINTER(lh_fr), _SZ, _SZ, C.POINTER(_P)]),
    "lh_zeromorph_vp_new": (C.c_int, [C.POINTER(lh_g1), C.POINTER(lh_g2), C.POINTER(lh_g2), C.POINTER(lh_g2),
                                      C.POINTER(_P)]),
    "lh_zeromorph_vp_export": (C.c_int, [_P, C.POINTER(lh_g1), C.POINTER(lh_g2), C.POINTER(lh_g2), C.POINTER(lh_g2)]),
    "lh_zeromorph_vp_free": (None, [_P]),
    "lh_zeromorph_verify": (C.c_int, [_P, C.POINTER(lh_g1), C.POINTER(lh_fr), _SZ, C.POINTER(lh_fr),
                                      C.POINTER(lh_transcript)]),
    "lh_zeromorph_batch_verify": (C.c_int, [_P, _SZ, C.POINTER(lh_g1), _SZ, C.POINTER(lh_fr), _SZ,
                                            C.POINTER(lh_evaluation), _SZ, C.POINTER(lh_transcript)]),
    "lh_lasso_prove_zeromorph": (C.c_int, [_P, _P, _SZ, C.POINTER(lh_lasso_table), _SZ, C.POINTER(_P),
                                           C.POINTER(lh_transcript)]),
    "lh_lasso_verify_zeromorph": (C.c_int, [_P, C.POINTER(lh_lasso_table), _SZ, C.POINTER(lh_transcript)]),
    "lh_hyperplonk_prove_zeromorph": (C.c_int, [_P, _P, _SZ, C.POINTER(lh_hp_param), C.POINTER(C.POINTER(lh_fr)),
                                                C.POINTER(_P), C.POINTER(lh_transcript)]),
    "lh_hyperplonk_verify_zeromorph": (C.c_int, [_P, C.POINTER(lh_hp_vparam), C.POINTER(C.POINTER(lh_fr)),
                                                 C.POINTER(lh_transcript)]),
    "lh_hyperplonk_prove_phases_zeromorph": (C.c_int, [_P, _P, _SZ, C.POINTER(lh_hp_param), _SZ, C.POINTER(_SZ),
                                                       C.POINTER(_SZ), C.POINTER(C.POINTER(lh_fr)),
                                                       C.POINTER(lh_hp_circuit), C.POINTER(lh_transcript)]),
    "lh_hyperplonk_verify_phases_zeromorph": (C.c_int, [_P, C.POINTER(lh_hp_vparam), _SZ, C.POINTER(_SZ), C.POINTER(_SZ),
                                                        C.POINTER(C.POINTER(lh_fr)), C.POINTER(lh_transcript)]),
    "lh_debug_jit_source": (C.c_int, [C.POINTER(C.c_uint32), _SZ, C.c_uint32, C.c_uint32, C.c_int, C.c_char_p, _SZ,
                                      C.POINTER(_SZ)]),
    "lh_profile_enable": (C.c_int, [_P, C.c_int]),
    "lh_profile_read": (C.c_int, [_P, C.POINTER(lh_prof_rec), _SZ, C.POINTER(_SZ)]),
}

_lib = None


def load():
    """dlopen the in-tree library and attach the signatures; raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(no CPU fallback exists)" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
