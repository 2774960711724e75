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
//! lockstep with the reference's and `ffi.rs` in lockstep with the header.  The same mapping, compiled and
//! tested, exists in C++ (`host/erased_cells.hpp`) and Python (`python/erased_cells_hip`); see INTEGRATION.md.
pub mod ffi;

mod cell_type;
mod cell_value;
mod device;
mod device_buffer;
#[cfg(feature = "masked")]
mod device_mask;
#[path = "errors.rs"]
pub mod error;
pub mod fused;
#[cfg(feature = "masked")]
mod masked;
#[cfg(feature = "masked")]
mod sentinel;
pub mod sharded;

pub use cell_type::*;
pub use cell_value::*;
pub use device::{init, set_stream, stream};
pub use device_buffer::*;
#[cfg(feature = "masked")]
pub use device_mask::*;
#[cfg(feature = "masked")]
pub use masked::*;
#[cfg(feature = "masked")]
pub use sentinel::*;
use std::fmt::{Debug, Formatter};

/// `with_ct` is a callback style macro used to construct various implementations covering all [`CellType`]s.
///
/// It calls the passed identifier as a macro with two parameters: the cell type id (e.g. `UInt8`) and the cell
/// type primitive (e.g. `u8`), for each of the ten encodings, in discriminant order.
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

/// Operations common to buffers of [`CellValue`]s.
pub trait BufferOps {
    /// Construct a [`CellBuffer`] from a `Vec<T>`.
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self;

    /// Construct a [`CellBuffer`] of given `len` length and `ct` `CellType`, filled with the type's default value.
    fn with_defaults(len: usize, ct: CellType) -> Self;

    /// Create a buffer of size `len` with all values `value`.
    fn fill(len: usize, value: CellValue) -> Self;

    /// Fill a buffer of size `len` with values from a closure called with the current index.
    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T;

    /// Get the length of the buffer.
    fn len(&self) -> usize;

    /// Determine if the buffer has zero values in it.
    fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// Get the cell-type of the encoded value.
    fn cell_type(&self) -> CellType;

    /// Get the [`CellValue`] at index `idx`.
    ///
    /// # Panics
    /// Will panic if `index` >= `self.len()`.
    fn get(&self, index: usize) -> CellValue;

    /// Store `value` at position `idx`.
    ///
    /// Returns `Err(NarrowingError)` if `value.cell_type() != self.cell_type()` and overflow could occur.
    ///
    /// # Panics
    /// Will panic if `index` >= `self.len()`.
    fn put(&mut self, index: usize, value: CellValue) -> error::Result<()>;

    /// Create a new buffer whereby all [`CellValue`]s are converted to `cell_type`.
    ///
    /// Returns `Err(NarrowingError)` if `cell_type` is narrower than the buffer's cell type.
    fn convert(&self, cell_type: CellType) -> error::Result<Self>
    where
        Self: Sized;

    /// Compute the minimum and maximum values the buffer.
    fn min_max(&self) -> (CellValue, CellValue);

    /// Convert `self` into a `Vec<T>`.
    fn to_vec<T: CellEncoding>(self) -> error::Result<Vec<T>>;
}

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
