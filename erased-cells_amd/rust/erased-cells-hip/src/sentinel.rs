//! No-data sentinels: `NoData<T>` and `IsNodata` (the reference's src/masked/nodata.rs).  A sentinel is compared with
//! cells under the total order — bitwise for floats — on the host for single values (`is`) and by one kernel for whole
//! buffers (`MaskedCellBuffer::from_vec_with_nodata` -> `ec_mask_from_nodata`).
use crate::ffi::ec_value;
use crate::{CellEncoding, CellType, CellValue};

/// Encodes a no-data value for cells that should be considered invalid or masked-out of a result.
#[derive(Debug, Copy, Clone, PartialEq, Default)]
pub enum NoData<T: CellEncoding> {
    /// No cell is no-data.
    None,
    /// The cell type's conventional sentinel: the minimum of an integer type, the canonical quiet NaN of a float type.
    #[default]
    Default,
    /// This value is the sentinel.
    Value(T),
}

/// `NoData::Default` for `T`.
fn conventional_sentinel<T: CellEncoding>() -> Option<T> {
    match T::cell_type() {
        CellType::Float32 => T::static_cast(f32::NAN),
        CellType::Float64 => T::static_cast(f64::NAN),
        integral => integral.min_value().get::<T>().ok(),
    }
}

impl<T: CellEncoding> NoData<T> {
    pub fn new(value: T) -> Self {
        NoData::Value(value)
    }
    pub fn value(&self) -> Option<T> {
        match *self {
            NoData::None => None,
            NoData::Value(v) => Some(v),
            NoData::Default => conventional_sentinel::<T>(),
        }
    }
    /// Determines if `value` should be considered a "no-data" value.  Equality is the total order's: a NaN sentinel
    /// matches only a NaN with the same bits, and -0.0 does not match +0.0; a value of another cell type is unified
    /// with the sentinel first.
    pub fn is(&self, value: &CellValue) -> bool {
        self.value().map_or(false, |sentinel| sentinel.into_cell_value() == *value)
    }
    /// The sentinel as the ABI's tagged scalar (`None` for `NoData::None`, which the ABI takes as a null pointer).
    pub(crate) fn to_ffi(&self) -> Option<ec_value> {
        self.value().map(|sentinel| sentinel.into_cell_value().to_ffi())
    }
}

/// Trait for no-data testing.
pub trait IsNodata {
    /// Determines if the `self` matches given `NoData` value.
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool;
}

impl IsNodata for CellValue {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        NoData::is(&no_data, self)
    }
}

impl<T: CellEncoding> IsNodata for T {
    fn is<N: CellEncoding>(&self, no_data: NoData<N>) -> bool {
        NoData::is(&no_data, &CellValue::new(*self))
    }
}
