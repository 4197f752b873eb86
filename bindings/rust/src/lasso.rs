//! The Lasso lookup argument (no counterpart in the reference snapshot; specification oracle/pyref/lasso.py and
//! oracle/pyref/hyperplonk.py LassoLookup).  NEVER COMPILED - see README.md.
use crate::{device::*, pcs::*, sys::*, transcript};
use halo2_curves::bn256::{Bn256, Fr, G1Affine};
use halo2_curves::ff::Field;
use plonkish_backend::{pcs::multilinear::MultilinearKzgVerifierParams, util::transcript::{TranscriptRead, TranscriptWrite}, Error};

fn table(num_chunks: u32, chunk_bits: u32, kind: u32, shift_bits: u32) -> lh_lasso_table {
    let mut t: lh_lasso_table = unsafe { std::mem::zeroed() };
    t.num_chunks = num_chunks;
    t.chunk_bits = chunk_bits;
    t.num_memories = num_chunks;
    t.num_terms = num_chunks;
    for j in 0..num_chunks as usize {
        t.memory_chunk[j] = j as u32;
        t.memory_subtable[j] = kind;
        // g = sum_j 2^(shift_bits * j) * E_j
        t.g_coeff[j] = Fr::from(2).pow([(shift_bits as u64) * j as u64]);
        t.g_num_factors[j] = 1;
        t.g_factor[j][0] = j as u8;
    }
    t
}
/// value < 2^(c*l): limbs looked up in the identity subtable (oracle/pyref/lasso.py range_table)
pub fn range_table(num_chunks: u32, chunk_bits: u32) -> lh_lasso_table {
    table(num_chunks, chunk_bits, LH_SUBTABLE_IDENTITY, chunk_bits)
}
/// AND / XOR of two (c*l/2)-bit operands, chunk j = x_j || y_j (oracle/pyref/lasso.py bitwise_table)
pub fn bitwise_table(xor: bool, num_chunks: u32, chunk_bits: u32) -> lh_lasso_table {
    table(num_chunks, chunk_bits, if xor { LH_SUBTABLE_XOR } else { LH_SUBTABLE_AND }, chunk_bits / 2)
}

/// dims[j]: chunk indices (< 2^chunk_bits) of every lookup, 2^num_vars entries each
pub fn prove(pp: &HipProverParam, table: &lh_lasso_table, num_vars: usize, dims: &[DeviceVec<u32>],
             transcript: &mut impl TranscriptWrite<G1Affine, Fr>) -> Result<(), Error> {
    let ptrs: Vec<*const u32> = dims.iter().map(|d| d.as_ptr()).collect();
    let mut vt = transcript::writer(transcript);
    check(unsafe { lh_lasso_prove(pp.ctx().raw(), pp.srs(), table, num_vars, ptrs.as_ptr(), &mut vt) })
}

pub fn verify(vp: &MultilinearKzgVerifierParams<Bn256>, table: &lh_lasso_table, num_vars: usize,
              transcript: &mut impl TranscriptRead<G1Affine, Fr>) -> Result<(), Error> {
    let h = vp_handle(vp)?;
    let mut vt = transcript::reader(transcript);
    check(unsafe { lh_lasso_verify(h.0, table, num_vars, &mut vt) })
}

/// A lookup of a HyperPlonk circuit proven by Lasso instead of LogUp: fill `lh_hp_param::lasso_lookups` /
/// `lh_hp_vparam::lasso_lookups` with these (backend.rs builds both structs; `PlonkishCircuitInfo` has no field for
/// it, so a circuit declares its Lasso lookups next to the info it hands to `preprocess`).
pub fn hp_lookup(table: lh_lasso_table, output_poly: usize, chunk_polys: &[usize]) -> lh_hp_lasso_lookup {
    let mut lk = lh_hp_lasso_lookup { table, output_poly, chunk_polys: [0; LH_LASSO_MAX_CHUNKS] };
    lk.chunk_polys[..chunk_polys.len()].copy_from_slice(chunk_polys);
    lk
}
