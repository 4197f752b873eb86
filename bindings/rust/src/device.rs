//! Device context, resident vectors, status mapping.  NEVER COMPILED - see README.md.
use crate::sys::*;
use halo2_curves::bn256::Fr;
use plonkish_backend::Error;
use std::{ffi::CStr, marker::PhantomData, os::raw::c_void, ptr, sync::Arc};

/// plonkish_backend::Error (lib.rs:12-20) from an lh_status; the message is the library's thread-local one.
pub fn check(rc: lh_status) -> Result<(), Error> {
    if rc == LH_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(lh_last_error()) }.to_string_lossy().into_owned();
    Err(match rc {
        LH_ERR_INVALID_SUMCHECK => Error::InvalidSumcheck(msg),
        LH_ERR_INVALID_PCS_PARAM => Error::InvalidPcsParam(msg),
        LH_ERR_INVALID_PCS_OPEN => Error::InvalidPcsOpen(msg),
        LH_ERR_INVALID_SNARK => Error::InvalidSnark(msg),
        LH_ERR_SERIALIZATION => Error::Serialization(msg),
        LH_ERR_TRANSCRIPT => Error::Transcript(std::io::ErrorKind::Other, msg),
        // no CPU fallback is attempted: a device failure or a violated precondition (an assert!/panic in the
        // reference) is a bug of the caller or the installation
        _ => panic!("liblasso_hip: {msg} (status {rc})"),
    })
}

struct CtxHandle(*mut lh_ctx);
unsafe impl Send for CtxHandle {}
unsafe impl Sync for CtxHandle {}
impl Drop for CtxHandle {
    fn drop(&mut self) {
        unsafe { lh_ctx_destroy(self.0) }
    }
}

/// One per GPU and host thread; calls on one context are not re-entrant (include/lasso_hip.h conventions).
#[derive(Clone)]
pub struct Context(Arc<CtxHandle>);

impl Context {
    pub fn new(device_id: i32) -> Result<Self, Error> {
        let mut p = ptr::null_mut();
        check(unsafe { lh_ctx_create(device_id, &mut p) })?;
        Ok(Context(Arc::new(CtxHandle(p))))
    }
    pub fn raw(&self) -> *mut lh_ctx {
        self.0 .0
    }
    pub fn upload<T: Copy>(&self, host: &[T]) -> Result<DeviceVec<T>, Error> {
        let bytes = std::mem::size_of_val(host);
        let mut d: *mut c_void = ptr::null_mut();
        check(unsafe { lh_alloc(self.raw(), bytes, &mut d) })?;
        check(unsafe { lh_upload(self.raw(), d, host.as_ptr() as *const c_void, bytes) })?;
        Ok(DeviceVec { ctx: self.clone(), ptr: d, len: host.len(), _t: PhantomData })
    }
    /// a `MultilinearPolynomial`'s evaluation table, resident (Fr is [u64; 4] Montgomery = lh_fr: no conversion)
    pub fn upload_frs(&self, evals: &[Fr]) -> Result<DeviceVec<Fr>, Error> {
        self.upload(evals)
    }
}

pub struct DeviceVec<T> {
    ctx: Context,
    ptr: *mut c_void,
    len: usize,
    _t: PhantomData<T>,
}
impl<T: Copy + Default> DeviceVec<T> {
    pub fn as_ptr(&self) -> *const T {
        self.ptr as *const T
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn download(&self) -> Result<Vec<T>, Error> {
        let mut out = vec![T::default(); self.len];
        check(unsafe {
            lh_download(self.ctx.raw(), out.as_mut_ptr() as *mut c_void, self.ptr, self.len * std::mem::size_of::<T>())
        })?;
        Ok(out)
    }
}
impl<T> Drop for DeviceVec<T> {
    fn drop(&mut self) {
        unsafe { lh_free(self.ctx.raw(), self.ptr) };
    }
}
