//! `SumCheck<Fr>` (piop/sum_check.rs:39-58) = ClassicSumCheck<EvaluationsProver> over a general VirtualPolynomial on the
//! GPU (rotations, Identity, Lagrange, eq_xy; classic.rs:208-240).  NEVER COMPILED - see README.md.
use crate::{device::*, expression::flatten, sys::*, transcript};
use halo2_curves::bn256::{Fr, G1Affine};
use plonkish_backend::{
    piop::sum_check::{SumCheck, VirtualPolynomial},
    util::transcript::{FieldTranscriptRead, FieldTranscriptWrite},
    Error,
};

#[derive(Clone, Debug)]
pub struct HipSumCheck;

/// FieldTranscript{Write,Read} carry no commitments: wrap them so that the callback table's commitment slots exist
/// (the sum-check never calls them)
mod field_only {
    use super::*;
    use std::os::raw::{c_int, c_void};
    pub unsafe extern "C" fn no_comm(_: *mut c_void, _: *const G1Affine) -> c_int {
        LH_ERR_TRANSCRIPT
    }
}

impl SumCheck<Fr> for HipSumCheck {
    type ProverParam = Context;
    type VerifierParam = ();

    fn prove(ctx: &Self::ProverParam, num_vars: usize, virtual_poly: VirtualPolynomial<Fr>, sum: Fr,
             transcript: &mut impl FieldTranscriptWrite<Fr>) -> Result<(Vec<Fr>, Vec<Fr>), Error> {
        // (VirtualPolynomial's fields are pub(crate) in the reference: the shim lives inside the crate, or the
        // reference grows four accessors)
        let (nodes, expr) = flatten(virtual_poly.expression());
        let resident: Vec<DeviceVec<Fr>> =
            virtual_poly.polys().iter().map(|p| ctx.upload_frs(p.evals())).collect::<Result<_, _>>()?;
        let ptrs: Vec<*const Fr> = resident.iter().map(|d| d.as_ptr()).collect();
        let ys: Vec<Fr> = virtual_poly.ys().iter().flat_map(|y| y.iter().copied()).collect();
        let mut vt = lh_transcript {
            user: transcript as *mut _ as *mut std::os::raw::c_void,
            write_field_element: Some(transcript::write_fe_of(transcript)),
            common_field_element: Some(transcript::common_fe_of(transcript)),
            squeeze_challenge: Some(transcript::squeeze_of(transcript)),
            write_commitment: Some(field_only::no_comm),
            common_commitment: Some(field_only::no_comm),
            read_field_element: None,
            read_commitment: None,
        };
        let mut x = vec![Fr::zero(); num_vars];
        let mut evals = vec![Fr::zero(); ptrs.len()];
        check(unsafe {
            lh_sumcheck_prove_expr(ctx.raw(), num_vars, &expr, ptrs.as_ptr(), ptrs.len(), virtual_poly.challenges().as_ptr(),
                                   virtual_poly.challenges().len(), ys.as_ptr(), virtual_poly.ys().len(), &sum, &mut vt,
                                   x.as_mut_ptr(), evals.as_mut_ptr())
        })?;
        drop(nodes);
        Ok((x, evals))
    }

    fn verify(_: &Self::VerifierParam, num_vars: usize, degree: usize, sum: Fr,
              transcript: &mut impl FieldTranscriptRead<Fr>) -> Result<(Fr, Vec<Fr>), Error> {
        // host only: classic.rs:242-272
        let mut vt = transcript::field_reader(transcript);
        let (mut eval, mut x) = (Fr::zero(), vec![Fr::zero(); num_vars]);
        check(unsafe { lh_sumcheck_verify(LH_SC_EVALUATIONS, num_vars, degree, &sum, &mut vt, &mut eval, x.as_mut_ptr()) })?;
        Ok((eval, x))
    }
}
