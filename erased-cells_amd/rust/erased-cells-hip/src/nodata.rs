//! `NoData<T>` and `IsNodata` (src/masked/nodata.rs of the reference).
use crate::{CellEncoding, CellType, CellValue};

/// Encodes a no-data value for cells that should be considered invalid or masked-out of a result.
#[derive(Debug, Copy, Clone, PartialEq, Default)]
pub enum NoData<T: CellEncoding> {
    /// Case where there is no no-data value.
    None,
    /// Case where there the default no-data value should be used: `T::MIN` for integers, the canonical NaN for floats.
    #[default]
    Default,
    /// Case where a specific no-data value is specified.
    Value(T),
}

impl<T: CellEncoding> NoData<T> {
    pub fn new(value: T) -> Self {
        NoData::Value(value)
    }
    pub fn value(&self) -> Option<T> {
        match self {
            NoData::None => None,
            NoData::Value(v) => Some(*v),
            NoData::Default => match T::cell_type() {
                CellType::Float32 => T::static_cast(<f32>::NAN),
                CellType::Float64 => T::static_cast(<f64>::NAN),
                ct => ct.min_value().get::<T>().ok(),
            },
        }
    }
    /// Determines if `value` should be considered a "no-data" value: equality under the total order, i.e. bitwise
    /// for floats (only a NaN with the same bits matches a NaN no-data value; −0.0 does not match +0.0).
    pub fn is(&self, value: &CellValue) -> bool {
        if let Some(nd_val) = self.value() {
            let nd_val2 = nd_val.into_cell_value();
            &nd_val2 == value
        } else {
            false
        }
    }
    /// The no-data value as the ABI's tagged scalar (None for `NoData::None`).
    pub(crate) fn to_ffi(&self) -> Option<crate::ffi::ec_value> {
        self.value().map(|v| v.into_cell_value().to_ffi())
    }
}

/// Trait for no-data testing.
pub trait IsNodata {
    /// Determines if the `self` matches given `NoData` value.
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool;
}

impl IsNodata for CellValue {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        no_data.is(self)
    }
}

impl<T: CellEncoding> IsNodata for T {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        no_data.is(&self.into_cell_value())
    }
}
