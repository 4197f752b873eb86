//! `PlonkishBackend<Fr>` (backend.rs:16-44) = HyperPlonk over multilinear KZG with `prove` on the GPU.
//! NEVER COMPILED - see README.md.
//!
//! setup / preprocess are the reference's (hyperplonk.rs:85-162: compose, permutation polys, preprocess commitments);
//! the preprocess and permutation polys are then uploaded once.  `prove` (hyperplonk.rs:164-291) is ONE call:
//! instances -> phase loop (circuit.synthesize is the callback of lh_hp_circuit) -> LogUp / permutation polys ->
//! zero-check -> evaluations -> batch_open, everything field-sized on the device.  `verify` (hyperplonk.rs:293-362)
//! runs in the host half of the library.
use crate::{device::*, expression::flatten, pcs::*, sys::*, transcript};
use halo2_curves::bn256::{Fr, G1Affine};
use plonkish_backend::{
    backend::{hyperplonk::{HyperPlonk, HyperPlonkProverParam, HyperPlonkVerifierParam}, PlonkishBackend, PlonkishCircuit,
              PlonkishCircuitInfo},
    pcs::PolynomialCommitmentScheme,
    util::transcript::{TranscriptRead, TranscriptWrite},
    Error,
};
use rand::RngCore;
use std::os::raw::{c_int, c_void};

#[derive(Clone, Debug)]
pub struct HipHyperPlonk;

/// device-resident half of HyperPlonkProverParam (hyperplonk.rs:38-55)
pub struct Resident {
    pub preprocess: Vec<DeviceVec<Fr>>,
    pub permutation: Vec<DeviceVec<Fr>>,
}

/// what `circuit.synthesize` hands back during one prove: kept alive until the call returns
struct SynthState<'a, C: PlonkishCircuit<Fr>> {
    circuit: &'a C,
    ctx: Context,
    alive: Vec<DeviceVec<Fr>>,
    error: Option<Error>,
}

unsafe extern "C" fn synthesize<C: PlonkishCircuit<Fr>>(user: *mut c_void, round: usize, challenges: *const Fr,
                                                         num_challenges: usize, d_out: *mut *const c_void,
                                                         num_out: usize) -> c_int {
    let st = &mut *(user as *mut SynthState<C>);
    let ch = std::slice::from_raw_parts(challenges, num_challenges);
    let polys = match st.circuit.synthesize(round, ch) {
        Ok(p) => p,
        Err(e) => {
            st.error = Some(e);
            return LH_ERR_INVALID_SNARK;
        }
    };
    if polys.len() != num_out {
        return LH_ERR_ARG; // assert_eq!(polys.len(), *num_witness_polys), hyperplonk.rs:198
    }
    for (i, p) in polys.iter().enumerate() {
        match st.ctx.upload_frs(p) {
            Ok(d) => {
                *d_out.add(i) = d.as_ptr() as *const c_void;
                st.alive.push(d);
            }
            Err(e) => {
                st.error = Some(e);
                return LH_ERR_DEVICE;
            }
        }
    }
    LH_OK
}

impl HipHyperPlonk {
    pub fn upload(pp: &HyperPlonkProverParam<Fr, HipMultilinearKzg>) -> Result<Resident, Error> {
        let ctx = pp.pcs.ctx();
        Ok(Resident {
            preprocess: pp.preprocess_polys.iter().map(|p| ctx.upload_frs(p.evals())).collect::<Result<_, _>>()?,
            permutation: pp.permutation_polys.iter().map(|(_, p)| ctx.upload_frs(p.evals())).collect::<Result<_, _>>()?,
        })
    }

    /// `prove` with the resident polys supplied (avoids re-uploading them per proof)
    pub fn prove_resident(pp: &HyperPlonkProverParam<Fr, HipMultilinearKzg>, resident: &Resident,
                          circuit: &impl PlonkishCircuit<Fr>, transcript: &mut impl TranscriptWrite<G1Affine, Fr>)
        -> Result<(), Error> {
        let ctx = pp.pcs.ctx().clone();
        let (nodes, expression) = flatten(&pp.expression);
        let lookup_nodes: Vec<Vec<(Vec<lh_expr_node>, lh_expr, Vec<lh_expr_node>, lh_expr)>> = pp.lookups.iter()
            .map(|lk| lk.iter().map(|(i, t)| { let (a, b) = flatten(i); let (c, d) = flatten(t); (a, b, c, d) }).collect())
            .collect();
        let inputs: Vec<Vec<lh_expr>> = lookup_nodes.iter().map(|lk| lk.iter().map(|x| x.1).collect()).collect();
        let tables: Vec<Vec<lh_expr>> = lookup_nodes.iter().map(|lk| lk.iter().map(|x| x.3).collect()).collect();
        let lookups: Vec<lh_hp_lookup> = inputs.iter().zip(&tables)
            .map(|(i, t)| lh_hp_lookup { inputs: i.as_ptr(), tables: t.as_ptr(), width: i.len() }).collect();
        let pre: Vec<*const c_void> = resident.preprocess.iter().map(|d| d.as_ptr() as *const c_void).collect();
        let perm: Vec<*const c_void> = resident.permutation.iter().map(|d| d.as_ptr() as *const c_void).collect();
        let perm_idx: Vec<usize> = pp.permutation_polys.iter().map(|(i, _)| *i).collect();
        let prm = lh_hp_param {
            num_vars: pp.num_vars,
            num_instance_polys: pp.num_instances.len(),
            num_instances: pp.num_instances.as_ptr(),
            num_preprocess_polys: pre.len(),
            d_preprocess_polys: pre.as_ptr(),
            num_witness_polys: pp.num_witness_polys.iter().sum(),
            num_challenges: pp.num_challenges.iter().sum(),
            num_lookups: lookups.len(),
            lookups: lookups.as_ptr(),
            num_permutation_polys: perm.len(),
            permutation_poly_index: perm_idx.as_ptr(),
            d_permutation_polys: perm.as_ptr(),
            num_permutation_z_polys: pp.num_permutation_z_polys,
            expression,
            num_lasso_lookups: 0, // see lasso.rs for circuits whose lookups are proven by Lasso
            lasso_lookups: std::ptr::null(),
        };
        let instances: Vec<*const Fr> = circuit.instances().iter().map(|i| i.as_ptr()).collect();
        let mut st = SynthState { circuit, ctx: ctx.clone(), alive: vec![], error: None };
        let circ = lh_hp_circuit { user: &mut st as *mut _ as *mut c_void, synthesize: Some(synthesize::<_>) };
        let mut vt = transcript::writer(transcript);
        let rc = unsafe {
            lh_hyperplonk_prove_phases(ctx.raw(), pp.pcs.srs(), &prm, pp.num_witness_polys.len(), pp.num_witness_polys.as_ptr(),
                                       pp.num_challenges.as_ptr(), instances.as_ptr(), &circ, &mut vt)
        };
        drop(nodes);
        if let Some(e) = st.error.take() {
            return Err(e); // the circuit's own error, not the library's summary of it
        }
        check(rc)
    }
}

impl PlonkishBackend<Fr> for HipHyperPlonk {
    type Pcs = HipMultilinearKzg;
    type ProverParam = HyperPlonkProverParam<Fr, HipMultilinearKzg>;
    type VerifierParam = HyperPlonkVerifierParam<Fr, HipMultilinearKzg>;

    fn setup(circuit_info: &PlonkishCircuitInfo<Fr>, rng: impl RngCore)
        -> Result<<Self::Pcs as PolynomialCommitmentScheme<Fr>>::Param, Error> {
        HyperPlonk::<HipMultilinearKzg>::setup(circuit_info, rng)
    }

    fn preprocess(param: &<Self::Pcs as PolynomialCommitmentScheme<Fr>>::Param, circuit_info: &PlonkishCircuitInfo<Fr>)
        -> Result<(Self::ProverParam, Self::VerifierParam), Error> {
        // the reference's preprocess is generic over the PCS: with HipMultilinearKzg its batch_commit calls already run
        // on the GPU
        HyperPlonk::<HipMultilinearKzg>::preprocess(param, circuit_info)
    }

    fn prove(pp: &Self::ProverParam, circuit: &impl PlonkishCircuit<Fr>,
             transcript: &mut impl TranscriptWrite<G1Affine, Fr>, _: impl RngCore) -> Result<(), Error> {
        let resident = Self::upload(pp)?; // callers proving repeatedly keep a `Resident` and call prove_resident
        Self::prove_resident(pp, &resident, circuit, transcript)
    }

    fn verify(vp: &Self::VerifierParam, instances: &[Vec<Fr>], transcript: &mut impl TranscriptRead<G1Affine, Fr>,
              _: impl RngCore) -> Result<(), Error> {
        let h = vp_handle(&vp.pcs)?;
        let (nodes, expression) = flatten(&vp.expression);
        let pre: Vec<G1Affine> = vp.preprocess_comms.iter().map(|c| c.0).collect();
        let perm: Vec<G1Affine> = vp.permutation_comms.iter().map(|(_, c)| c.0).collect();
        let prm = lh_hp_vparam {
            num_vars: vp.num_vars,
            num_instance_polys: vp.num_instances.len(),
            num_instances: vp.num_instances.as_ptr(),
            num_witness_polys: vp.num_witness_polys.iter().sum(),
            num_challenges: vp.num_challenges.iter().sum(),
            num_lookups: vp.num_lookups,
            num_permutation_z_polys: vp.num_permutation_z_polys,
            expression,
            num_preprocess_polys: pre.len(),
            preprocess_comms: pre.as_ptr(),
            num_permutation_polys: perm.len(),
            permutation_comms: perm.as_ptr(),
            num_lasso_lookups: 0,
            lasso_lookups: std::ptr::null(),
        };
        let inst: Vec<*const Fr> = instances.iter().map(|i| i.as_ptr()).collect();
        let mut vt = transcript::reader(transcript);
        let rc = unsafe {
            lh_hyperplonk_verify_phases(h.0, &prm, vp.num_witness_polys.len(), vp.num_witness_polys.as_ptr(),
                                        vp.num_challenges.as_ptr(), inst.as_ptr(), &mut vt)
        };
        drop(nodes);
        check(rc)
    }
}
