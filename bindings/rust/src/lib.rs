//! `plonkish_backend` traits on top of `liblasso_hip.so` (include/lasso_hip.h).  NEVER COMPILED - see README.md.
pub mod backend;
pub mod device;
pub mod expression;
pub mod lasso;
pub mod pcs;
pub mod sum_check;
pub mod sys;
pub mod transcript;

pub use backend::HipHyperPlonk;
pub use device::{Context, DeviceVec};
pub use pcs::HipMultilinearKzg;
pub use sum_check::HipSumCheck;
