//! `PolynomialCommitmentScheme<Fr>` (pcs.rs:22-130) for multilinear KZG over bn256 with commit / open on the GPU.
//! NEVER COMPILED - see README.md.
//!
//! Param / VerifierParam / Commitment types are the reference's own (`MultilinearKzgParams`, `..VerifierParams`,
//! `MultilinearKzgCommitment`, pcs/multilinear/kzg.rs:25-117): proofs and parameter files stay interchangeable.
//! The prover param adds the device copy of `eqs` (flat: level k at offset 2^k - 1, exactly `lh_srs_upload`'s layout).
use crate::{device::*, sys::*, transcript};
use halo2_curves::bn256::{Bn256, Fr, G1Affine, G2Affine};
use plonkish_backend::{
    pcs::{
        multilinear::{MultilinearKzg, MultilinearKzgCommitment, MultilinearKzgParams, MultilinearKzgProverParams,
                      MultilinearKzgVerifierParams},
        Evaluation, Point, PolynomialCommitmentScheme,
    },
    poly::multilinear::MultilinearPolynomial,
    util::transcript::{TranscriptRead, TranscriptWrite},
    Error,
};
use rand::RngCore;
use serde::{Deserialize, Serialize};
use std::{ptr, sync::Arc};

#[derive(Clone, Debug)]
pub struct HipMultilinearKzg;

struct SrsHandle(Context, *mut lh_srs);
unsafe impl Send for SrsHandle {}
unsafe impl Sync for SrsHandle {}
impl Drop for SrsHandle {
    fn drop(&mut self) {
        unsafe { lh_srs_free(self.0.raw(), self.1) }
    }
}

/// the reference's prover param plus its device-resident copy (not serialized: re-uploaded by `trim` / on first use)
#[derive(Clone, Serialize, Deserialize)]
pub struct HipProverParam {
    pub host: MultilinearKzgProverParams<Bn256>,
    #[serde(skip)]
    device: Option<Arc<SrsHandle>>,
}
impl std::fmt::Debug for HipProverParam {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        self.host.fmt(f)
    }
}

impl HipProverParam {
    pub fn upload(host: MultilinearKzgProverParams<Bn256>, ctx: &Context) -> Result<Self, Error> {
        let flat: Vec<G1Affine> = host.eqs().iter().flat_map(|lvl| lvl.iter().copied()).collect();
        let mut srs = ptr::null_mut();
        check(unsafe { lh_srs_upload(ctx.raw(), flat.as_ptr(), host.num_vars(), &mut srs) })?;
        Ok(Self { host, device: Some(Arc::new(SrsHandle(ctx.clone(), srs))) })
    }
    pub fn ctx(&self) -> &Context {
        &self.device.as_ref().expect("prover param not uploaded").0
    }
    pub fn srs(&self) -> *const lh_srs {
        self.device.as_ref().expect("prover param not uploaded").1
    }
}

/// verifier side: the host library's copy of (g1, g2, ss)
pub struct VpHandle(pub *mut lh_mkzg_vp);
unsafe impl Send for VpHandle {}
unsafe impl Sync for VpHandle {}
impl Drop for VpHandle {
    fn drop(&mut self) {
        unsafe { lh_mkzg_vp_free(self.0) }
    }
}
pub fn vp_handle(vp: &MultilinearKzgVerifierParams<Bn256>) -> Result<VpHandle, Error> {
    let (g1, g2): (G1Affine, G2Affine) = (vp.g1(), vp.g2());
    let ss = vp.ss(vp.num_vars());
    let mut out = ptr::null_mut();
    check(unsafe { lh_mkzg_vp_new(&g1, &g2, ss.as_ptr(), ss.len(), &mut out) })?;
    Ok(VpHandle(out))
}

fn evaluations(evals: &[Evaluation<Fr>]) -> Vec<lh_evaluation> {
    evals.iter().map(|e| lh_evaluation { poly: e.poly() as u32, point: e.point() as u32, value: *e.value() }).collect()
}

impl PolynomialCommitmentScheme<Fr> for HipMultilinearKzg {
    type Param = MultilinearKzgParams<Bn256>;
    type ProverParam = HipProverParam;
    type VerifierParam = MultilinearKzgVerifierParams<Bn256>;
    type Polynomial = MultilinearPolynomial<Fr>;
    type Commitment = MultilinearKzgCommitment<G1Affine>;
    type CommitmentChunk = G1Affine;

    // setup: the reference's (kzg.rs:166-228; it draws the trapdoor from `rng`) - a one-off, not on the hot path
    fn setup(poly_size: usize, batch_size: usize, rng: impl RngCore) -> Result<Self::Param, Error> {
        MultilinearKzg::<Bn256>::setup(poly_size, batch_size, rng)
    }

    // trim (kzg.rs:230-250), then the bases go to the GPU once
    fn trim(param: &Self::Param, poly_size: usize, batch_size: usize) -> Result<(Self::ProverParam, Self::VerifierParam), Error> {
        let (pp, vp) = MultilinearKzg::<Bn256>::trim(param, poly_size, batch_size)?;
        Ok((HipProverParam::upload(pp, &Context::new(0)?)?, vp))
    }

    fn commit(pp: &Self::ProverParam, poly: &Self::Polynomial) -> Result<Self::Commitment, Error> {
        Ok(Self::batch_commit(pp, [poly])?.pop().unwrap())
    }

    // kzg.rs:259-274: one batched MSM on the device
    fn batch_commit<'a>(pp: &Self::ProverParam, polys: impl IntoIterator<Item = &'a Self::Polynomial>)
        -> Result<Vec<Self::Commitment>, Error> {
        let polys: Vec<_> = polys.into_iter().collect();
        if polys.is_empty() {
            return Ok(vec![]);
        }
        let resident: Vec<DeviceVec<Fr>> = polys.iter().map(|p| pp.ctx().upload_frs(p.evals())).collect::<Result<_, _>>()?;
        let ptrs: Vec<*const Fr> = resident.iter().map(|d| d.as_ptr()).collect();
        let mut out = vec![G1Affine::default(); polys.len()];
        check(unsafe {
            lh_mkzg_batch_commit(pp.ctx().raw(), pp.srs(), ptrs.as_ptr(), ptrs.len(), polys[0].num_vars(), out.as_mut_ptr())
        })?;
        Ok(out.into_iter().map(MultilinearKzgCommitment).collect())
    }

    // kzg.rs:276-302: quotients and their commitments on the device, written to the caller's transcript
    fn open(pp: &Self::ProverParam, poly: &Self::Polynomial, _comm: &Self::Commitment, point: &Point<Fr, Self::Polynomial>,
            _eval: &Fr, transcript: &mut impl TranscriptWrite<G1Affine, Fr>) -> Result<(), Error> {
        let d = pp.ctx().upload_frs(poly.evals())?;
        let mut vt = transcript::writer(transcript);
        let mut eval = Fr::zero();
        check(unsafe { lh_mkzg_open(pp.ctx().raw(), pp.srs(), d.as_ptr(), poly.num_vars(), point.as_ptr(), &mut vt, &mut eval) })
    }

    // pcs/multilinear.rs:134-235 (additive::batch_open): merge, sum-check, open - all behind one call
    fn batch_open<'a>(pp: &Self::ProverParam, polys: impl IntoIterator<Item = &'a Self::Polynomial>,
                      _comms: impl IntoIterator<Item = &'a Self::Commitment>, points: &[Point<Fr, Self::Polynomial>],
                      evals: &[Evaluation<Fr>], transcript: &mut impl TranscriptWrite<G1Affine, Fr>) -> Result<(), Error> {
        let polys: Vec<_> = polys.into_iter().collect();
        let num_vars = polys[0].num_vars();
        let resident: Vec<DeviceVec<Fr>> = polys.iter().map(|p| pp.ctx().upload_frs(p.evals())).collect::<Result<_, _>>()?;
        let ptrs: Vec<*const Fr> = resident.iter().map(|d| d.as_ptr()).collect();
        let flat: Vec<Fr> = points.iter().flat_map(|p| p.iter().copied()).collect();
        let evs = evaluations(evals);
        let mut vt = transcript::writer(transcript);
        check(unsafe {
            lh_mkzg_batch_open(pp.ctx().raw(), pp.srs(), num_vars, ptrs.as_ptr(), ptrs.len(), flat.as_ptr(), points.len(),
                               evs.as_ptr(), evs.len(), &mut vt)
        })
    }

    fn read_commitments(vp: &Self::VerifierParam, num_polys: usize, transcript: &mut impl TranscriptRead<G1Affine, Fr>)
        -> Result<Vec<Self::Commitment>, Error> {
        MultilinearKzg::<Bn256>::read_commitments(vp, num_polys, transcript)
    }

    // kzg.rs:330-361 / pcs/multilinear.rs:237-276 in the host half of the library (BN254 optimal ate)
    fn verify(vp: &Self::VerifierParam, comm: &Self::Commitment, point: &Point<Fr, Self::Polynomial>, eval: &Fr,
              transcript: &mut impl TranscriptRead<G1Affine, Fr>) -> Result<(), Error> {
        let h = vp_handle(vp)?;
        let mut vt = transcript::reader(transcript);
        check(unsafe { lh_mkzg_verify(h.0, &comm.0, point.as_ptr(), point.len(), eval, &mut vt) })
    }

    fn batch_verify<'a>(vp: &Self::VerifierParam, comms: impl IntoIterator<Item = &'a Self::Commitment>,
                        points: &[Point<Fr, Self::Polynomial>], evals: &[Evaluation<Fr>],
                        transcript: &mut impl TranscriptRead<G1Affine, Fr>) -> Result<(), Error> {
        let h = vp_handle(vp)?;
        let comms: Vec<G1Affine> = comms.into_iter().map(|c| c.0).collect();
        let flat: Vec<Fr> = points.iter().flat_map(|p| p.iter().copied()).collect();
        let evs = evaluations(evals);
        let mut vt = transcript::reader(transcript);
        check(unsafe {
            lh_mkzg_batch_verify(h.0, points[0].len(), comms.as_ptr(), comms.len(), flat.as_ptr(), points.len(),
                                 evs.as_ptr(), evs.len(), &mut vt)
        })
    }
}
