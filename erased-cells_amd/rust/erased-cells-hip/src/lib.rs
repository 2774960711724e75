//! `erased_cells` with the per-cell arithmetic on an MI355X: device-resident `CellBuffer` / `Mask` /
//! `MaskedCellBuffer` whose operator bodies are calls into liberased_cells_hip.so (C ABI: include/erased_cells.h).
//!
//! The public surface is the reference's (erased-cells 0.1.1): the same three core enums-by-name [`CellType`],
//! [`CellValue`], [`CellBuffer`], the [`BufferOps`] and [`CellEncoding`] traits, the `with_ct!` macro, and — with
//! the `masked` feature — [`Mask`], [`MaskedCellBuffer`], [`NoData`], `IsNodata`.  `CellType` keeps its
//! discriminants (they ARE the ABI dtype codes); the host keeps the type tag, zip truncation, the
//! empty-result-is-UInt8 rule and the length asserts; each per-cell loop body of the reference becomes one FFI
//! call on HBM-resident cells.  What differs: `CellBuffer` is a struct (tag + length + device block) instead of
//! an enum over `Vec<T>`, so code that matches on `CellBuffer::UInt8(vec)` must call `to_vec::<u8>()` instead;
//! `Error` has one more variant (`Backend`).  Extras: [`fused`] (operator chains in one pass), [`sharded`]
//! (row-block shards over the GPUs of a node), [`init`] / [`set_stream`].
//!
//! NOT COMPILED in the build image (no rustc there); `tests/test_rust_surface.py` keeps the public items in
//! lockstep with the reference's and `ffi.rs` in lockstep with the header, parameter types included.  The same
//! mapping, compiled and tested, exists in C++ (`host/erased_cells.hpp`) and Python (`python/erased_cells_hip`).
//!
//! PROVENANCE.  This crate is a derived work of erased-cells 0.1.1 (MIT License, Copyright (c) 2023 Astraea, Inc.):
//! its public item declarations — names, signatures, trait and operator impl headers, the `with_ct!` table — are the
//! reference's, kept verbatim where a drop-in needs them and marked `api-surface(<reference file>:<lines>)` in the
//! source; everything between and inside them is this crate's own.  INTEGRATION.md §2 lists every marked range;
//! `tests/test_rust_provenance.py` holds the list and the sources together.
pub mod ffi;

mod cell_type;
mod cell_value;
mod encoding;
mod device;
mod device_buffer;
#[cfg(feature = "masked")]
mod device_mask;
pub mod error;
pub mod fused;
#[cfg(feature = "masked")]
mod masked_buffer;
#[cfg(feature = "masked")]
mod nodata;
pub mod sharded;

pub use cell_type::*;
pub use cell_value::*;
pub use encoding::*;
pub use device::{init, set_stream, stream};
pub use device_buffer::*;
#[cfg(feature = "masked")]
pub use device_mask::*;
#[cfg(feature = "masked")]
pub use masked_buffer::*;
#[cfg(feature = "masked")]
pub use nodata::*;
use std::fmt::{Debug, Formatter};

// api-surface(src/lib.rs:60-101): the `with_ct!` table
/// Callback macro over the ten cell encodings: `with_ct!(m)` expands to `m! { (UInt8, u8), (UInt16, u16), ... (Float64, f64) }`
/// — the pairs of `CellType` variant and Rust primitive in discriminant order (which is also the ABI's dtype code order).
/// This is how the crate, like the reference, stamps out anything that must exist once per cell type.
#[macro_export]
macro_rules! with_ct {
    ($callback:ident) => {
        $callback! {
            (UInt8, u8),
            (UInt16, u16),
            (UInt32, u32),
            (UInt64, u64),
            (Int8, i8),
            (Int16, i16),
            (Int32, i32),
            (Int64, i64),
            (Float32, f32),
            (Float64, f64)
        }
    };
}
// end api-surface

// api-surface(src/lib.rs:103-163): trait BufferOps (the method set and signatures; the doc comments are this crate's)
/// What [`CellBuffer`] and [`MaskedCellBuffer`] have in common.  Every method that touches cells is a call into the
/// library on HBM-resident data; what each costs is noted where it differs from a host `Vec`.
pub trait BufferOps {
    /// Upload `data` (one host-to-HBM copy); the cell type is `T`'s.
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self;

    /// `len` cells of type `ct`, all zero (a fill kernel; nothing is uploaded).
    fn with_defaults(len: usize, ct: CellType) -> Self;

    /// `len` copies of `value`; the buffer takes `value`'s cell type.
    fn fill(len: usize, value: CellValue) -> Self;

    /// Cell `i` is `f(i)`: evaluated on the host, uploaded once.
    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T;

    /// Number of cells.
    fn len(&self) -> usize;

    /// `len() == 0`.
    fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// The cell type every cell of the buffer has.
    fn cell_type(&self) -> CellType;

    /// One cell, downloaded (a device round trip: iterate or `to_vec` to read many).
    ///
    /// # Panics
    /// If `index >= self.len()`.
    fn get(&self, index: usize) -> CellValue;

    /// Overwrite one cell.  `value` is widened to the buffer's cell type first; `Err(NarrowingError)` (and nothing
    /// written) if its own cell type does not fit.
    ///
    /// # Panics
    /// If `index >= self.len()`.
    fn put(&mut self, index: usize, value: CellValue) -> error::Result<()>;

    /// A new buffer with every cell widened to `cell_type` (one kernel); the same type gives a device-to-device copy.
    /// `Err(NarrowingError)` before any device work when `cell_type` cannot hold the buffer's type.
    fn convert(&self, cell_type: CellType) -> error::Result<Self>
    where
        Self: Sized;

    /// `(min, max)` under the total order (integers by value, floats by `total_cmp`), typed as the buffer: a device
    /// reduction.  Folded from `(T::MAX, T::MIN)`, so an empty buffer returns that inverted pair.
    fn min_max(&self) -> (CellValue, CellValue);

    /// Widen to `T` on the device if needed, then download (`Err(NarrowingError)` if `T` is narrower).
    fn to_vec<T: CellEncoding>(self) -> error::Result<Vec<T>>;
}
// end api-surface

/// Newtype wrapper for debug rendering: more than ten items show as the first five, `, ... `, the last five.
pub(crate) struct Elided<'a, T>(&'a [T]);

impl<T: Debug> Debug for Elided<'_, T> {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        let v = self.0;
        let show = |f: &mut Formatter<'_>, part: &[T]| -> std::fmt::Result {
            for (i, x) in part.iter().enumerate() {
                if i > 0 {
                    f.write_str(", ")?;
                }
                f.write_fmt(format_args!("{x:?}"))?;
            }
            Ok(())
        };
        if v.len() > 10 {
            show(f, &v[..5])?;
            f.write_str(", ... ")?;
            show(f, &v[v.len() - 5..])
        } else {
            show(f, v)
        }
    }
}
