//! The lh_transcript callback table over the reference's transcript traits (util/transcript.rs:15-97).
//! NEVER COMPILED - see README.md.
use crate::sys::*;
use halo2_curves::bn256::{Fr, G1Affine};
use plonkish_backend::util::transcript::{
    FieldTranscript, FieldTranscriptRead, FieldTranscriptWrite, Transcript, TranscriptRead, TranscriptWrite,
};
use std::os::raw::{c_int, c_void};

fn status<T>(r: Result<T, plonkish_backend::Error>) -> (c_int, Option<T>) {
    match r {
        Ok(v) => (LH_OK, Some(v)),
        Err(_) => (LH_ERR_TRANSCRIPT, None),
    }
}

unsafe extern "C" fn write_fe<T: FieldTranscriptWrite<Fr>>(u: *mut c_void, fe: *const Fr) -> c_int {
    status((*(u as *mut T)).write_field_element(&*fe)).0
}
unsafe extern "C" fn common_fe<T: FieldTranscript<Fr>>(u: *mut c_void, fe: *const Fr) -> c_int {
    status((*(u as *mut T)).common_field_element(&*fe)).0
}
unsafe extern "C" fn squeeze<T: FieldTranscript<Fr>>(u: *mut c_void, out: *mut Fr) -> c_int {
    *out = (*(u as *mut T)).squeeze_challenge();
    LH_OK
}
unsafe extern "C" fn write_comm<T: TranscriptWrite<G1Affine, Fr>>(u: *mut c_void, pt: *const G1Affine) -> c_int {
    status((*(u as *mut T)).write_commitment(&*pt)).0
}
unsafe extern "C" fn common_comm<T: Transcript<G1Affine, Fr>>(u: *mut c_void, pt: *const G1Affine) -> c_int {
    status((*(u as *mut T)).common_commitment(&*pt)).0
}
unsafe extern "C" fn read_fe<T: FieldTranscriptRead<Fr>>(u: *mut c_void, out: *mut Fr) -> c_int {
    let (rc, v) = status((*(u as *mut T)).read_field_element());
    if let Some(v) = v {
        *out = v;
    }
    rc
}
unsafe extern "C" fn read_comm<T: TranscriptRead<G1Affine, Fr>>(u: *mut c_void, out: *mut G1Affine) -> c_int {
    let (rc, v) = status((*(u as *mut T)).read_commitment());
    if let Some(v) = v {
        *out = v;
    }
    rc
}

/// `&mut impl TranscriptWrite<G1Affine, Fr>` as the library sees it; valid while `t` is borrowed
pub fn writer<T: TranscriptWrite<G1Affine, Fr>>(t: &mut T) -> lh_transcript {
    lh_transcript {
        user: t as *mut T as *mut c_void,
        write_field_element: Some(write_fe::<T>),
        common_field_element: Some(common_fe::<T>),
        squeeze_challenge: Some(squeeze::<T>),
        write_commitment: Some(write_comm::<T>),
        common_commitment: Some(common_comm::<T>),
        read_field_element: None,
        read_commitment: None,
    }
}

unsafe extern "C" fn no_write_fe(_: *mut c_void, _: *const Fr) -> c_int {
    LH_ERR_TRANSCRIPT // a reading transcript is never written to
}
unsafe extern "C" fn no_write_comm(_: *mut c_void, _: *const G1Affine) -> c_int {
    LH_ERR_TRANSCRIPT
}

/// `&mut impl TranscriptRead<G1Affine, Fr>` for the verifiers: read_* plus the shared common_* / squeeze of the base
/// traits (util/transcript.rs:15-43); the write slots hold stubs that fail
pub fn reader<T: TranscriptRead<G1Affine, Fr>>(t: &mut T) -> lh_transcript {
    lh_transcript {
        user: t as *mut T as *mut c_void,
        write_field_element: Some(no_write_fe),
        common_field_element: Some(common_fe::<T>),
        squeeze_challenge: Some(squeeze::<T>),
        write_commitment: Some(no_write_comm),
        common_commitment: Some(common_comm::<T>),
        read_field_element: Some(read_fe::<T>),
        read_commitment: Some(read_comm::<T>),
    }
}

// ---- helpers for the field-only transcripts of SumCheck::{prove, verify} (piop/sum_check.rs:43-57)
pub fn write_fe_of<T: FieldTranscriptWrite<Fr>>(_: &T) -> FeCb {
    write_fe::<T>
}
pub fn common_fe_of<T: FieldTranscript<Fr>>(_: &T) -> FeCb {
    common_fe::<T>
}
pub fn squeeze_of<T: FieldTranscript<Fr>>(_: &T) -> FeOutCb {
    squeeze::<T>
}
pub fn field_reader<T: FieldTranscriptRead<Fr>>(t: &mut T) -> lh_transcript {
    lh_transcript {
        user: t as *mut T as *mut c_void,
        write_field_element: Some(no_write_fe),
        common_field_element: Some(common_fe::<T>),
        squeeze_challenge: Some(squeeze::<T>),
        write_commitment: Some(no_write_comm),
        common_commitment: Some(no_write_comm),
        read_field_element: Some(read_fe::<T>),
        read_commitment: None,
    }
}
