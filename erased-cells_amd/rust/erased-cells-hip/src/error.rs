//! `Result` / `Error` of the crate.
//!
//! PROVENANCE.  The `Error` enum — its variants and their messages, which callers may match on or print — is the
//! reference's (erased-cells 0.1.1, src/error.rs:9-27, MIT License, Copyright (c) 2023 Astraea, Inc.) plus one variant
//! of this crate's for failures of the HIP backend, which a CPU crate cannot have.  See INTEGRATION.md §2.
use crate::ffi::*;
use crate::CellType;
use std::ffi::CStr;

// api-surface(src/error.rs:9-27): the Result alias and the Error enum with the reference's five variants and messages
pub type Result<T, E = Error> = std::result::Result<T, E>;

#[derive(thiserror::Error, Debug)]
pub enum Error {
    /// Widening only: `convert`, `put`, `get::<T>` and `to_vec::<T>` refuse a narrower target type up front.
    #[error("Invalid narrowing from cell-type {src} to {dst}")]
    NarrowingError { src: CellType, dst: CellType },
    #[error("Unsupported cell-type {0}")]
    UnsupportedCellTypeError(String),
    #[error("Expected a value but received `None`: {0}")]
    ExpectedError(String),
    #[error("Unable to parse {0} as a {1}")]
    ParseError(String, &'static str),
    #[error("Unable to convert {0} into NoData<{1}>::Value")]
    NoDataConversionError(f64, &'static str),
    /// An `ec_status` other than `EC_OK` / `EC_ERR_NARROWING` (no device, out of HBM, RCCL): `ec_last_error_string()`.
    #[error("HIP backend: {0}")]
    Backend(String),
}
// end api-surface

/// `ec_status` -> `Result`: `EC_ERR_NARROWING` becomes `Error::NarrowingError { src, dst }` again.
pub(crate) fn check(st: ec_status) -> Result<()> {
    if st == EC_OK {
        return Ok(());
    }
    if st == EC_ERR_NARROWING {
        let (mut s, mut d) = (0u8, 0u8);
        unsafe { ec_last_narrowing(&mut s, &mut d) };
        return Err(Error::NarrowingError { src: CellType::from_code(s), dst: CellType::from_code(d) });
    }
    Err(Error::Backend(unsafe { CStr::from_ptr(ec_last_error_string()) }.to_string_lossy().into_owned()))
}

/// Arithmetic is infallible in the reference; a backend failure (no device, out of HBM) is a panic.
pub(crate) fn must(st: ec_status, what: &str) {
    if let Err(e) = check(st) {
        panic!("{what}: {e}");
    }
}
