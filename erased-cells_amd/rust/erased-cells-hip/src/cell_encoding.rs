//! `CellEncoding` (src/encoding.rs:9-40 of the reference): the ten primitives a cell can hold.
use crate::{with_ct, CellType, CellValue};
use num_traits::{One, Zero};
use std::fmt::Debug;

/// Trait for marking Rust primitives as having a corresponding [`CellType`].
pub trait CellEncoding: Copy + Debug + Default + Zero + One + PartialEq {
    /// Returns the [`CellType`] covering `Self`.
    fn cell_type() -> CellType;
    /// Converts `self` into a [`CellValue`].
    fn into_cell_value(self) -> CellValue;
    /// Convert dynamic type to static type when logically known: `None` unless `T` is exactly `Self`.
    fn static_cast<T: CellEncoding + Sized>(value: T) -> Option<Self> {
        if Self::cell_type() == T::cell_type() {
            // same cell type => same primitive: a bit copy
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
