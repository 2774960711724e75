//! `CellEncoding`: which Rust primitives can be a cell.
//!
//! PROVENANCE.  The trait declaration and the `encoding!` stamp below are the reference's public surface
//! (erased-cells 0.1.1, src/encoding.rs:9-40, MIT License, Copyright (c) 2023 Astraea, Inc.) and are kept as they are so
//! that `impl CellEncoding`-bounded code written against the reference compiles against this crate.  See INTEGRATION.md §2.
use crate::{with_ct, CellType, CellValue};
use num_traits::{One, Zero};
use std::fmt::Debug;

// api-surface(src/encoding.rs:9-40): trait CellEncoding and its ten impls
/// A Rust primitive that has a [`CellType`] of its own: exactly `u8 u16 u32 u64 i8 i16 i32 i64 f32 f64`.
pub trait CellEncoding: Copy + Debug + Default + Zero + One + PartialEq {
    /// The [`CellType`] whose cells are `Self`.
    fn cell_type() -> CellType;
    /// `self` as a [`CellValue`] of that cell type.
    fn into_cell_value(self) -> CellValue;
    /// Recover the static type of a value whose dynamic cell type is known: `Some` exactly when `T` IS `Self`
    /// (a bit-for-bit copy), `None` for every other pair.  No numeric conversion happens here — that is
    /// `CellValue::convert`.
    fn static_cast<T: CellEncoding + Sized>(value: T) -> Option<Self> {
        if Self::cell_type() == T::cell_type() {
            Some(unsafe { std::mem::transmute_copy::<T, Self>(&value) })
        } else {
            None
        }
    }
}

macro_rules! encoding {
    ( $( ($ct:ident, $prim:ident) ),* ) => { $(
        impl CellEncoding for $prim {
            fn cell_type() -> CellType {
                CellType::$ct
            }
            fn into_cell_value(self) -> CellValue {
                CellValue::$ct(self)
            }
        } )*
    };
}

with_ct!(encoding);
// end api-surface
