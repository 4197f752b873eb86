//! `extern "C"` declarations of include/lasso_hip.h.  NEVER COMPILED - see README.md.
//! Field elements and points cross by pointer cast: lh_fr = bn256::Fr, lh_g1 = bn256::G1Affine, lh_g2 = bn256::G2Affine.
#![allow(non_camel_case_types, dead_code)]
use halo2_curves::bn256::{Fr, G1Affine, G2Affine};
use std::os::raw::{c_char, c_int, c_void};

pub type lh_status = c_int;
pub const LH_OK: lh_status = 0;
pub const LH_ERR_INVALID_SUMCHECK: lh_status = -1;
pub const LH_ERR_INVALID_PCS_PARAM: lh_status = -2;
pub const LH_ERR_INVALID_PCS_OPEN: lh_status = -3;
pub const LH_ERR_INVALID_SNARK: lh_status = -4;
pub const LH_ERR_SERIALIZATION: lh_status = -5;
pub const LH_ERR_TRANSCRIPT: lh_status = -6;
pub const LH_ERR_DEVICE: lh_status = -7;
pub const LH_ERR_ARG: lh_status = -8;

pub const LH_SC_MAX_TERMS: usize = 48;
pub const LH_SC_MAX_FACTORS: usize = 4;
pub const LH_SC_EVALUATIONS: c_int = 0;
pub const LH_SC_COEFFICIENTS: c_int = 1;
pub const LH_LASSO_MAX_CHUNKS: usize = 8;
pub const LH_LASSO_MAX_MEMORIES: usize = 16;
pub const LH_LASSO_MAX_TERMS: usize = 16;
pub const LH_RCCL_UNIQUE_ID_BYTES: usize = 128;

#[repr(C)] pub struct lh_ctx { _p: [u8; 0] }
#[repr(C)] pub struct lh_srs { _p: [u8; 0] }
#[repr(C)] pub struct lh_usrs { _p: [u8; 0] }
#[repr(C)] pub struct lh_mkzg_vp { _p: [u8; 0] }
#[repr(C)] pub struct lh_zm_vp { _p: [u8; 0] }

pub type FeCb = unsafe extern "C" fn(*mut c_void, *const Fr) -> c_int;
pub type FeOutCb = unsafe extern "C" fn(*mut c_void, *mut Fr) -> c_int;
pub type G1Cb = unsafe extern "C" fn(*mut c_void, *const G1Affine) -> c_int;
pub type G1OutCb = unsafe extern "C" fn(*mut c_void, *mut G1Affine) -> c_int;

#[repr(C)]
pub struct lh_transcript {
    pub user: *mut c_void,
    pub write_field_element: Option<FeCb>,
    pub common_field_element: Option<FeCb>,
    pub squeeze_challenge: Option<FeOutCb>,
    pub write_commitment: Option<G1Cb>,
    pub common_commitment: Option<G1Cb>,
    pub read_field_element: Option<FeOutCb>,
    pub read_commitment: Option<G1OutCb>,
}

#[repr(C)]
pub struct lh_sop {
    pub num_terms: u32,
    pub global_eq: i32,
    pub coeff: [Fr; LH_SC_MAX_TERMS],
    pub num_factors: [u8; LH_SC_MAX_TERMS],
    pub factor: [[u8; LH_SC_MAX_FACTORS]; LH_SC_MAX_TERMS],
}

pub const LH_EX_CONSTANT: u32 = 0;
pub const LH_EX_IDENTITY: u32 = 1;
pub const LH_EX_LAGRANGE: u32 = 2;
pub const LH_EX_EQ_XY: u32 = 3;
pub const LH_EX_POLYNOMIAL: u32 = 4;
pub const LH_EX_CHALLENGE: u32 = 5;
pub const LH_EX_NEGATED: u32 = 6;
pub const LH_EX_SUM: u32 = 7;
pub const LH_EX_PRODUCT: u32 = 8;
pub const LH_EX_SCALED: u32 = 9;
#[repr(C)] #[derive(Clone, Copy)]
pub struct lh_expr_node { pub op: u32, pub a: i32, pub b: i32, pub reserved: u32, pub scalar: Fr }
#[repr(C)] #[derive(Clone, Copy)]
pub struct lh_expr { pub nodes: *const lh_expr_node, pub num_nodes: usize }

#[repr(C)] #[derive(Clone, Copy)]
pub struct lh_evaluation { pub poly: u32, pub point: u32, pub value: Fr }

pub const LH_SUBTABLE_IDENTITY: u32 = 0;
pub const LH_SUBTABLE_AND: u32 = 1;
pub const LH_SUBTABLE_XOR: u32 = 2;
#[repr(C)] #[derive(Clone, Copy)]
pub struct lh_lasso_table {
    pub num_chunks: u32,
    pub chunk_bits: u32,
    pub num_memories: u32,
    pub memory_chunk: [u32; LH_LASSO_MAX_MEMORIES],
    pub memory_subtable: [u32; LH_LASSO_MAX_MEMORIES],
    pub num_terms: u32,
    pub g_coeff: [Fr; LH_LASSO_MAX_TERMS],
    pub g_num_factors: [u8; LH_LASSO_MAX_TERMS],
    pub g_factor: [[u8; LH_SC_MAX_FACTORS]; LH_LASSO_MAX_TERMS],
}

#[repr(C)]
pub struct lh_comm {
    pub rank: c_int,
    pub size: c_int,
    pub user: *mut c_void,
    pub all_gather: Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize) -> c_int>,
    pub all_gather_device: Option<unsafe extern "C" fn(*mut c_void, *const c_void, *mut c_void, usize, *mut c_void) -> c_int>,
}

#[repr(C)]
pub struct lh_hp_lookup { pub inputs: *const lh_expr, pub tables: *const lh_expr, pub width: usize }
#[repr(C)] #[derive(Clone, Copy)]
pub struct lh_hp_lasso_lookup {
    pub table: lh_lasso_table,
    pub output_poly: usize,
    pub chunk_polys: [usize; LH_LASSO_MAX_CHUNKS],
}
#[repr(C)]
pub struct lh_hp_param {
    pub num_vars: usize,
    pub num_instance_polys: usize,
    pub num_instances: *const usize,
    pub num_preprocess_polys: usize,
    pub d_preprocess_polys: *const *const c_void,
    pub num_witness_polys: usize,
    pub num_challenges: usize,
    pub num_lookups: usize,
    pub lookups: *const lh_hp_lookup,
    pub num_permutation_polys: usize,
    pub permutation_poly_index: *const usize,
    pub d_permutation_polys: *const *const c_void,
    pub num_permutation_z_polys: usize,
    pub expression: lh_expr,
    pub num_lasso_lookups: usize,
    pub lasso_lookups: *const lh_hp_lasso_lookup,
}
#[repr(C)]
pub struct lh_hp_vparam {
    pub num_vars: usize,
    pub num_instance_polys: usize,
    pub num_instances: *const usize,
    pub num_witness_polys: usize,
    pub num_challenges: usize,
    pub num_lookups: usize,
    pub num_permutation_z_polys: usize,
    pub expression: lh_expr,
    pub num_preprocess_polys: usize,
    pub preprocess_comms: *const G1Affine,
    pub num_permutation_polys: usize,
    pub permutation_comms: *const G1Affine,
    pub num_lasso_lookups: usize,
    pub lasso_lookups: *const lh_hp_lasso_lookup,
}
#[repr(C)]
pub struct lh_hp_circuit {
    pub user: *mut c_void,
    pub synthesize: Option<unsafe extern "C" fn(*mut c_void, usize, *const Fr, usize, *mut *const c_void, usize) -> c_int>,
}

/// lh_lasso_route (include/lasso_hip.h): which routes the last Lasso prove on a ctx took
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct lh_lasso_route {
    pub open_small_depth: u32,
    pub open_small_passes: u32,
    pub eq_factored_rounds: u32,
    pub standard_rounds: u32,
    pub rw_leaf_rounds: u32,
    pub resident_tails: u32,
    pub resident_rounds: u32,
    pub packed_ts_pairs: u32,
    pub derived_commitments: u32,
    pub sorted_dim_reuse: u32,
    pub sharded_rounds: u32,
    pub shard_exchanges: u32,
    pub window_table_jobs: u32,
    pub open_precommit: u32,
    pub resident_layers: u32,
    pub pp_folds: u32,
    pub msm_half_batches: u32,
    pub reserved: [u32; 7],
}

extern "C" {
    pub fn lh_last_error() -> *const c_char;
    pub fn lh_version() -> *const c_char;
    // context & memory
    pub fn lh_ctx_create(device_id: c_int, out: *mut *mut lh_ctx) -> lh_status;
    pub fn lh_ctx_destroy(ctx: *mut lh_ctx);
    pub fn lh_ctx_sync(ctx: *mut lh_ctx) -> lh_status;
    pub fn lh_ctx_stream(ctx: *mut lh_ctx) -> *mut c_void;
    pub fn lh_alloc(ctx: *mut lh_ctx, bytes: usize, d_out: *mut *mut c_void) -> lh_status;
    pub fn lh_free(ctx: *mut lh_ctx, d_ptr: *mut c_void) -> lh_status;
    pub fn lh_upload(ctx: *mut lh_ctx, d_dst: *mut c_void, src: *const c_void, bytes: usize) -> lh_status;
    pub fn lh_download(ctx: *mut lh_ctx, dst: *mut c_void, d_src: *const c_void, bytes: usize) -> lh_status;
    // Fr vectors / MultilinearPolynomial
    pub fn lh_fr_from_u64(ctx: *mut lh_ctx, d_in: *const u64, n: usize, d_out: *mut Fr) -> lh_status;
    pub fn lh_fr_from_u32(ctx: *mut lh_ctx, d_in: *const u32, n: usize, d_out: *mut Fr) -> lh_status;
    pub fn lh_fr_batch_invert(ctx: *mut lh_ctx, d_in: *const Fr, n: usize, d_out: *mut Fr) -> lh_status;
    pub fn lh_fix_var(ctx: *mut lh_ctx, d_in: *const Fr, n_in: usize, x: *const Fr, d_out: *mut Fr) -> lh_status;
    pub fn lh_eq_xy(ctx: *mut lh_ctx, y: *const Fr, num_vars: usize, d_out: *mut Fr) -> lh_status;
    pub fn lh_evaluate(ctx: *mut lh_ctx, d_polys: *const *const Fr, num_polys: usize, num_vars: usize,
                       point: *const Fr, out_evals: *mut Fr) -> lh_status;
    pub fn lh_lincomb(ctx: *mut lh_ctx, d_polys: *const *const Fr, w: *const Fr, num_polys: usize, n: usize,
                      d_out: *mut Fr) -> lh_status;
    // piop::sum_check, piop::gkr
    pub fn lh_sumcheck_prove(ctx: *mut lh_ctx, prover_kind: c_int, num_vars: usize, expr: *const lh_sop,
                             d_polys: *const *const Fr, num_polys: usize, ys: *const Fr, num_ys: usize,
                             sum: *const Fr, t: *mut lh_transcript, out_challenges: *mut Fr, out_evals: *mut Fr) -> lh_status;
    pub fn lh_sumcheck_prove_expr(ctx: *mut lh_ctx, num_vars: usize, expr: *const lh_expr, d_polys: *const *const Fr,
                                  num_polys: usize, challenges: *const Fr, num_challenges: usize, ys: *const Fr,
                                  num_ys: usize, sum: *const Fr, t: *mut lh_transcript, out_challenges: *mut Fr,
                                  out_evals: *mut Fr) -> lh_status;
    pub fn lh_sumcheck_verify(prover_kind: c_int, num_vars: usize, degree: usize, sum: *const Fr,
                              t: *mut lh_transcript, out_eval: *mut Fr, out_x: *mut Fr) -> lh_status;
    pub fn lh_gkr_fractional_prove(ctx: *mut lh_ctx, num_batching: usize, num_vars: usize,
                                   claimed_p_0s: *const *const Fr, claimed_q_0s: *const *const Fr,
                                   d_ps: *const *const Fr, d_qs: *const *const Fr, t: *mut lh_transcript,
                                   out_p_xs: *mut Fr, out_q_xs: *mut Fr, out_x: *mut Fr) -> lh_status;
    // util::arithmetic::msm
    pub fn lh_msm(ctx: *mut lh_ctx, d_scalars: *const Fr, d_bases: *const G1Affine, n: usize, out: *mut G1Affine) -> lh_status;
    pub fn lh_msm_u32(ctx: *mut lh_ctx, d_scalars: *const u32, d_bases: *const G1Affine, n: usize, out: *mut G1Affine) -> lh_status;
    // pcs::multilinear::kzg
    pub fn lh_srs_upload(ctx: *mut lh_ctx, eqs_flat: *const G1Affine, num_vars: usize, out: *mut *mut lh_srs) -> lh_status;
    pub fn lh_srs_num_vars(srs: *const lh_srs) -> usize;
    pub fn lh_srs_free(ctx: *mut lh_ctx, srs: *mut lh_srs);
    pub fn lh_mkzg_batch_commit(ctx: *mut lh_ctx, srs: *const lh_srs, d_polys: *const *const Fr, num_polys: usize,
                                num_vars: usize, out_comms: *mut G1Affine) -> lh_status;
    pub fn lh_mkzg_open(ctx: *mut lh_ctx, srs: *const lh_srs, d_poly: *const Fr, num_vars: usize, point: *const Fr,
                        t: *mut lh_transcript, out_eval: *mut Fr) -> lh_status;
    pub fn lh_mkzg_batch_open(ctx: *mut lh_ctx, srs: *const lh_srs, num_vars: usize, d_polys: *const *const Fr,
                              num_polys: usize, points: *const Fr, num_points: usize, evals: *const lh_evaluation,
                              num_evals: usize, t: *mut lh_transcript) -> lh_status;
    pub fn lh_mkzg_vp_new(g1: *const G1Affine, g2: *const G2Affine, ss: *const G2Affine, num_vars: usize,
                          out: *mut *mut lh_mkzg_vp) -> lh_status;
    pub fn lh_mkzg_vp_free(vp: *mut lh_mkzg_vp);
    pub fn lh_mkzg_verify(vp: *const lh_mkzg_vp, comm: *const G1Affine, point: *const Fr, num_vars: usize,
                          eval: *const Fr, t: *mut lh_transcript) -> lh_status;
    pub fn lh_mkzg_batch_verify(vp: *const lh_mkzg_vp, num_vars: usize, comms: *const G1Affine, num_comms: usize,
                                points: *const Fr, num_points: usize, evals: *const lh_evaluation, num_evals: usize,
                                t: *mut lh_transcript) -> lh_status;
    // Lasso
    pub fn lh_lasso_prove(ctx: *mut lh_ctx, srs: *const lh_srs, table: *const lh_lasso_table, num_vars: usize,
                          d_dims: *const *const u32, t: *mut lh_transcript) -> lh_status;
    pub fn lh_lasso_verify(vp: *const lh_mkzg_vp, table: *const lh_lasso_table, num_vars: usize,
                           t: *mut lh_transcript) -> lh_status;
    // one proof over several GPUs
    pub fn lh_ctx_set_comm(ctx: *mut lh_ctx, comm: *const lh_comm, shard_bit: usize) -> lh_status;
    pub fn lh_rccl_unique_id(out: *mut u8) -> lh_status;
    pub fn lh_ctx_set_comm_rccl(ctx: *mut lh_ctx, rank: c_int, size: c_int, unique_id: *const u8, shard_bit: usize) -> lh_status;
    pub fn lh_ctx_set_comm_loopback(ctx: *mut lh_ctx, rank: c_int, size: c_int, shard_bit: usize) -> lh_status;
    pub fn lh_ctx_comm_stats(ctx: *mut lh_ctx, out: *mut u64) -> lh_status;
    pub fn lh_ctx_comm_phase_stats(ctx: *mut lh_ctx, out: *mut u64, reset: c_int) -> lh_status;
    pub fn lh_ctx_memory_stats(ctx: *mut lh_ctx, out: *mut u64) -> lh_status;
    pub fn lh_ctx_host_cpus(ctx: *mut lh_ctx, bus_id: *mut c_char, bus_id_cap: usize, cpulist: *mut c_char, cpulist_cap: usize) -> lh_status;
    pub fn lh_shard_extract(ctx: *mut lh_ctx, d_global: *const c_void, n_local: usize, shard_bit: usize, rho: usize,
                            rank: usize, elem_bytes: usize, d_local: *mut c_void) -> lh_status;
    // route options (include/lasso_hip.h lists the names) and the route the last Lasso prove took
    pub fn lh_ctx_set_option(ctx: *mut lh_ctx, name: *const c_char, value: i64) -> lh_status;
    pub fn lh_ctx_get_option(ctx: *mut lh_ctx, name: *const c_char, out: *mut i64) -> lh_status;
    pub fn lh_lasso_last_route(ctx: *mut lh_ctx, out: *mut lh_lasso_route) -> lh_status;
    pub fn lh_lasso_prove_sharded(ctx: *mut lh_ctx, srs: *const lh_srs, table: *const lh_lasso_table, num_vars: usize,
                                  d_dims_local: *const *const u32, t: *mut lh_transcript) -> lh_status;
    // HyperPlonk
    pub fn lh_hyperplonk_prove(ctx: *mut lh_ctx, srs: *const lh_srs, pp: *const lh_hp_param,
                               instances: *const *const Fr, d_witness_polys: *const *const Fr,
                               t: *mut lh_transcript) -> lh_status;
    pub fn lh_hyperplonk_prove_sharded(ctx: *mut lh_ctx, srs: *const lh_srs, pp: *const lh_hp_param,
                                       instances: *const *const Fr, d_witness_polys_local: *const *const Fr,
                                       t: *mut lh_transcript) -> lh_status;
    pub fn lh_hyperplonk_prove_phases(ctx: *mut lh_ctx, srs: *const lh_srs, pp: *const lh_hp_param, num_phases: usize,
                                      num_witness_polys: *const usize, num_challenges: *const usize,
                                      instances: *const *const Fr, circuit: *const lh_hp_circuit,
                                      t: *mut lh_transcript) -> lh_status;
    pub fn lh_hyperplonk_verify(vp: *const lh_mkzg_vp, hvp: *const lh_hp_vparam, instances: *const *const Fr,
                                t: *mut lh_transcript) -> lh_status;
    pub fn lh_hyperplonk_verify_phases(vp: *const lh_mkzg_vp, hvp: *const lh_hp_vparam, num_phases: usize,
                                       num_witness_polys: *const usize, num_challenges: *const usize,
                                       instances: *const *const Fr, t: *mut lh_transcript) -> lh_status;
    // multi-phase circuits over Zeromorph (hyperplonk.rs:185-205 is PCS-generic; lh_usrs / lh_zm_vp are opaque here)
    pub fn lh_hyperplonk_prove_phases_zeromorph(ctx: *mut lh_ctx, srs: *const core::ffi::c_void, poly_size: usize,
                                                pp: *const lh_hp_param, num_phases: usize,
                                                num_witness_polys: *const usize, num_challenges: *const usize,
                                                instances: *const *const Fr, circuit: *const lh_hp_circuit,
                                                t: *mut lh_transcript) -> lh_status;
    pub fn lh_hyperplonk_verify_phases_zeromorph(vp: *const core::ffi::c_void, hvp: *const lh_hp_vparam, num_phases: usize,
                                                 num_witness_polys: *const usize, num_challenges: *const usize,
                                                 instances: *const *const Fr, t: *mut lh_transcript) -> lh_status;
    // (development) the source text of a runtime-compiled round kernel
    pub fn lh_debug_jit_source(code: *const u32, num_instrs: usize, num_regs: u32, result_reg: u32, degree: i32,
                               out: *mut core::ffi::c_char, cap: usize, len: *mut usize) -> lh_status;
    // Zeromorph over univariate KZG: lh_ukzg_setup, lh_usrs_*, lh_zeromorph_* follow the same shapes
    // (include/lasso_hip.h, section f3) and are bound the same way when HyperPlonk<Zeromorph<..>> is wanted.
}
