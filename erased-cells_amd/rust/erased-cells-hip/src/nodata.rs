//! No-data sentinels: [`NoData<T>`] and [`IsNodata`].
//!
//! PROVENANCE.  The enum, the method signatures and the `IsNodata` trait with its two impl headers are the reference's
//! public surface (erased-cells 0.1.1, src/masked/nodata.rs:7-68, MIT License, Copyright (c) 2023 Astraea, Inc.); the
//! bodies are this crate's.  See INTEGRATION.md §2.
//!
//! A sentinel is compared with cells under the total order — bitwise for floats — on the host for single values
//! (`is`) and by one kernel for whole buffers (`MaskedCellBuffer::from_vec_with_nodata` -> `ec_mask_from_nodata`).
use crate::ffi::ec_value;
use crate::{CellEncoding, CellValue};

// api-surface(src/masked/nodata.rs:7-17): the enum
/// How a buffer marks cells that hold no measurement.
#[derive(Debug, Copy, Clone, PartialEq, Default)]
pub enum NoData<T: CellEncoding> {
    /// Every cell is data.
    None,
    /// The cell type's conventional marker: `T::MIN` for the integer types, the canonical quiet NaN for the float types.
    #[default]
    Default,
    /// Cells equal to this value are no-data.
    Value(T),
}
// end api-surface

impl<T: CellEncoding> NoData<T> {
    pub fn new(value: T) -> Self {
        NoData::Value(value)
    }

    /// The marker as a `T`, if there is one.  `Default` asks the library's type table for the cell type's minimum
    /// (`ec_min_value`) and, for the two float types, substitutes the canonical NaN.
    pub fn value(&self) -> Option<T> {
        match *self {
            NoData::None => None,
            NoData::Value(v) => Some(v),
            NoData::Default => {
                let ct = T::cell_type();
                let marker = if ct.is_integral() {
                    ct.min_value()
                } else if ct.size_of() == 4 {
                    CellValue::Float32(f32::NAN)
                } else {
                    CellValue::Float64(f64::NAN)
                };
                marker.get::<T>().ok()
            }
        }
    }

    /// Is `value` the marker?  Equality is the total order's: a NaN marker matches only a NaN with the same bits,
    /// -0.0 does not match +0.0, and a value of another cell type is unified with the marker before comparing.
    pub fn is(&self, value: &CellValue) -> bool {
        match self.value() {
            Some(marker) => CellValue::new(marker).cmp(value).is_eq(),
            None => false,
        }
    }

    /// The marker as the ABI's tagged scalar; `None` (a null pointer across the ABI) for `NoData::None`.
    pub(crate) fn to_ffi(&self) -> Option<ec_value> {
        Some(CellValue::new(self.value()?).to_ffi())
    }
}

// api-surface(src/masked/nodata.rs:52-68): trait IsNodata and its impl headers
/// No-data test from the value's side: `cell.is(NoData::Default)`.
pub trait IsNodata {
    /// Does `self` equal the marker `no_data` stands for?
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool;
}

impl IsNodata for CellValue {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        NoData::is(&no_data, self)
    }
}

impl<T: CellEncoding> IsNodata for T {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        NoData::is(&no_data, &(*self).into())
    }
}
// end api-surface
