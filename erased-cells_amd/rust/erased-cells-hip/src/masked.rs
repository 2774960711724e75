//! `MaskedCellBuffer` (src/masked/masked_buffer.rs of the reference): a [`CellBuffer`] with a companion [`Mask`],
//! both device-resident.  The value op over ALL cells and the AND of the masks run as one launch.
use crate::device::stream;
use crate::error::{check, must};
use crate::ffi::*;
use crate::{BufferOps, CellBuffer, CellEncoding, CellType, CellValue, Mask, NoData};
use std::fmt::{Debug, Formatter};

/// A [`CellBuffer`] with a companion [`Mask`].
///
/// The `Mask` tracks which cells are valid across operations, and which should be treated as "no-data" values.
#[derive(Clone, PartialEq, PartialOrd)]
pub struct MaskedCellBuffer(CellBuffer, Mask);

impl MaskedCellBuffer {
    /// Create a new combined [`CellBuffer`] and [`Mask`].
    ///
    /// # Panics
    /// Will panics if `buffer` and `mask` are not the same length.
    pub fn new(buffer: CellBuffer, mask: Mask) -> Self {
        assert_eq!(buffer.len(), mask.len(), "Mask and buffer must have the same length.");
        Self(buffer, mask)
    }

    /// Constructs a `MaskedCellBuffer` from a `Vec<CellEncoding>`, specifying a `NoData<T>` value.
    ///
    /// Mask value will be `false` when associated cell matches `nodata` (one kernel over the uploaded cells;
    /// the reference walks the cells with `IsNodata::is`, which stays available for single values).
    pub fn from_vec_with_nodata<T: CellEncoding>(data: Vec<T>, nodata: NoData<T>) -> Self {
        let buf = CellBuffer::from_vec(data);
        let mask = Mask::uninit(buf.len());
        let nd = nodata.to_ffi();
        let nd_ptr = nd.as_ref().map_or(std::ptr::null(), |v| v as *const ec_value);
        must(
            unsafe { ec_mask_from_nodata(buf.ct as u8, buf.dev_ptr(), buf.len(), nd_ptr, mask.dev_ptr_mut(), stream()) },
            "ec_mask_from_nodata",
        );
        Self::new(buf, mask)
    }

    pub fn fill_with_mask_via<T, F>(len: usize, mv: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> (T, bool),
    {
        (0..len).map(mv).collect()
    }

    pub fn buffer(&self) -> &CellBuffer {
        &self.0
    }

    pub fn buffer_mut(&mut self) -> &mut CellBuffer {
        &mut self.0
    }

    pub fn mask(&self) -> &Mask {
        &self.1
    }

    pub fn mask_mut(&mut self) -> &mut Mask {
        &mut self.1
    }

    /// Get a buffer value at position `index` with mask evaluated.
    ///
    /// Returns `Some(CellValue)` if mask at `index` is `true`, `None` otherwise.
    pub fn get_masked(&self, index: usize) -> Option<CellValue> {
        if self.mask().get(index) {
            Some(self.buffer().get(index))
        } else {
            None
        }
    }

    /// Get the cell value and mask value at position `index`.
    pub fn get_with_mask(&self, index: usize) -> (CellValue, bool) {
        (self.buffer().get(index), self.mask().get(index))
    }

    /// Set the `value` and `mask` at position `index`.
    ///
    /// Returns `Err(NarrowingError)` if `value` cannot be converted to `self.cell_type()` without data loss.
    pub fn put_with_mask(&mut self, index: usize, value: CellValue, mask: bool) -> crate::error::Result<()> {
        self.put(index, value)?;
        self.mask_mut().put(index, mask);
        Ok(())
    }

    /// Returns a tuple of representing counts of `(data, nodata)`.
    pub fn counts(&self) -> (usize, usize) {
        self.mask().counts()
    }

    /// Convert `self` into a `Vec<T>`, replacing values where the mask is `0` to `no_data.value()`
    pub fn to_vec_with_nodata<T: CellEncoding>(self, no_data: NoData<T>) -> crate::error::Result<Vec<T>> {
        let Self(buf, mask) = self;
        let conv = buf.convert(T::cell_type())?;
        let nd = match no_data.to_ffi() {
            None => return conv.to_vec::<T>(),
            Some(v) => v,
        };
        assert_eq!(conv.cell_type(), T::cell_type()); // as `danger::cast` asserts (an empty convert yields UInt8)
        let sel = CellBuffer::uninit(conv.cell_type(), conv.len());
        check(unsafe { ec_mask_select(conv.ct as u8, conv.dev_ptr(), mask.dev_ptr(), conv.len(), &nd, sel.mem.ptr(), stream()) })?;
        sel.to_vec::<T>()
    }

    fn binop(&self, op: ec_op, rhs: &Self) -> Self {
        let n = self.len().min(rhs.len());
        if n == 0 {
            return Self(CellBuffer::empty_u8(), Mask::uninit(0));
        }
        // the buffer op over ALL cells and `lmask & rmask` in one launch (src/masked/masked_buffer.rs:326-335)
        let (out, om) = (CellBuffer::uninit(CellType::Float64, n), Mask::uninit(n));
        must(
            unsafe {
                ec_masked_binop(op, self.0.ct as u8, self.0.dev_ptr(), self.1.dev_ptr(), rhs.0.ct as u8, rhs.0.dev_ptr(),
                                rhs.1.dev_ptr(), n, out.mem.ptr() as *mut f64, om.dev_ptr_mut(), stream())
            },
            "ec_masked_binop",
        );
        Self(out, om)
    }
}

impl BufferOps for MaskedCellBuffer {
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        let buffer = CellBuffer::from_vec(data);
        let mask = Mask::fill(buffer.len(), true);
        Self::new(buffer, mask)
    }

    fn with_defaults(len: usize, ct: CellType) -> Self {
        let buffer = CellBuffer::with_defaults(len, ct);
        let mask = Mask::fill(len, true);
        Self::new(buffer, mask)
    }

    fn fill(len: usize, value: CellValue) -> Self {
        let buffer = CellBuffer::fill(len, value);
        let mask = Mask::fill(len, true);
        Self::new(buffer, mask)
    }

    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T,
    {
        let buffer = CellBuffer::fill_via(len, f);
        let mask = Mask::fill(len, true);
        Self::new(buffer, mask)
    }

    fn len(&self) -> usize {
        self.buffer().len()
    }

    fn cell_type(&self) -> CellType {
        self.buffer().cell_type()
    }

    fn get(&self, index: usize) -> CellValue {
        self.buffer().get(index)
    }

    fn put(&mut self, idx: usize, value: CellValue) -> crate::error::Result<()> {
        self.buffer_mut().put(idx, value)
    }

    fn convert(&self, cell_type: CellType) -> crate::error::Result<Self>
    where
        Self: Sized,
    {
        let converted = self.buffer().convert(cell_type)?;
        Ok(Self::new(converted, self.mask().to_owned()))
    }

    /// `min_max` restricted to the cells whose mask is `true` (all masked -> the inverted sentinels).
    fn min_max(&self) -> (CellValue, CellValue) {
        let (mut mn, mut mx) = (CellValue::UInt8(0).to_ffi(), CellValue::UInt8(0).to_ffi());
        must(
            unsafe { ec_min_max(self.0.ct as u8, self.0.dev_ptr(), self.1.dev_ptr(), self.len(), &mut mn, &mut mx, stream()) },
            "ec_min_max",
        );
        (CellValue::from_ffi(&mn), CellValue::from_ffi(&mx))
    }

    /// Converts `self` to `Vec<T>`, ignoring the `mask` values.
    fn to_vec<T: CellEncoding>(self) -> crate::error::Result<Vec<T>> {
        self.0.to_vec()
    }
}

impl Debug for MaskedCellBuffer {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        let basename = self.cell_type().to_string();
        f.debug_tuple(&format!("{basename}MaskedCellBuffer"))
            .field(self.buffer())
            .field(self.mask())
            .finish()
    }
}

impl From<MaskedCellBuffer> for (CellBuffer, Mask) {
    fn from(value: MaskedCellBuffer) -> Self {
        (value.0, value.1)
    }
}

impl<'a> From<&'a MaskedCellBuffer> for (&'a CellBuffer, &'a Mask) {
    fn from(value: &'a MaskedCellBuffer) -> Self {
        (&value.0, &value.1)
    }
}

/// Converts a [`CellBuffer`] into a [`MaskedCellBuffer`] with an all-true mask.
impl From<CellBuffer> for MaskedCellBuffer {
    fn from(value: CellBuffer) -> Self {
        let len = value.len();
        Self::new(value, Mask::fill(len, true))
    }
}

impl<C: CellEncoding> FromIterator<C> for MaskedCellBuffer {
    fn from_iter<T: IntoIterator<Item = C>>(iter: T) -> Self {
        Self::from_vec(iter.into_iter().collect())
    }
}

impl<C: CellEncoding> FromIterator<(C, bool)> for MaskedCellBuffer {
    /// The cell type is `C`'s, also for an empty iterator.
    fn from_iter<T: IntoIterator<Item = (C, bool)>>(iter: T) -> Self {
        let (data, mask): (Vec<C>, Vec<bool>) = iter.into_iter().unzip();
        Self::new(CellBuffer::from_vec(data), Mask::new(mask))
    }
}

impl<C: CellEncoding> Extend<(C, bool)> for MaskedCellBuffer {
    fn extend<T: IntoIterator<Item = (C, bool)>>(&mut self, iter: T) {
        let (data, mask): (Vec<C>, Vec<bool>) = iter.into_iter().unzip();
        self.buffer_mut().extend(data); // one device reallocation per call, not per item
        self.mask_mut().extend(mask);
    }
}

impl<'buf> IntoIterator for &'buf MaskedCellBuffer {
    type Item = (CellValue, bool);
    type IntoIter = MaskedCellBufferIterator<'buf>;

    fn into_iter(self) -> Self::IntoIter {
        MaskedCellBufferIterator { cells: self.buffer().into_iter(), mask: self.mask().to_vec().into_iter() }
    }
}

/// Iterator over ([`CellValue`], `bool`) elements in a [`MaskedCellBuffer`] (two downloads for the whole walk).
pub struct MaskedCellBufferIterator<'buf> {
    cells: crate::CellBufferIterator<'buf>,
    mask: std::vec::IntoIter<bool>,
}

impl Iterator for MaskedCellBufferIterator<'_> {
    type Item = (CellValue, bool);

    fn next(&mut self) -> Option<Self::Item> {
        match (self.cells.next(), self.mask.next()) {
            (Some(v), Some(m)) => Some((v, m)),
            _ => None,
        }
    }
}

mod ops {
    use crate::ffi::*;
    use crate::{CellValue, MaskedCellBuffer};
    use std::ops::{Add, Div, Mul, Neg, Sub};

    macro_rules! cb_bin_op {
        ($trt:ident, $mth:ident, $op:expr) => {
            // Both borrows.
            impl $trt for &MaskedCellBuffer {
                type Output = MaskedCellBuffer;
                fn $mth(self, rhs: Self) -> Self::Output {
                    self.binop($op, rhs)
                }
            }
            // Both owned/consumed
            impl $trt for MaskedCellBuffer {
                type Output = MaskedCellBuffer;
                fn $mth(self, rhs: Self) -> Self::Output {
                    $trt::$mth(&self, &rhs)
                }
            }
            // RHS borrow
            impl $trt<&MaskedCellBuffer> for MaskedCellBuffer {
                type Output = MaskedCellBuffer;
                fn $mth(self, rhs: &MaskedCellBuffer) -> Self::Output {
                    $trt::$mth(&self, rhs)
                }
            }
            // RHS scalar: the mask is moved over unchanged
            impl<R> $trt<R> for MaskedCellBuffer
            where
                R: Into<CellValue>,
            {
                type Output = MaskedCellBuffer;
                fn $mth(self, rhs: R) -> Self::Output {
                    let r: CellValue = rhs.into();
                    let (buf, mask) = self.into();
                    let new_buf = buf.binop_scalar($op, r);
                    Self::new(new_buf, mask)
                }
            }
        };
    }
    cb_bin_op!(Add, add, EC_ADD);
    cb_bin_op!(Sub, sub, EC_SUB);
    cb_bin_op!(Mul, mul, EC_MUL);
    cb_bin_op!(Div, div, EC_DIV);

    impl Neg for &MaskedCellBuffer {
        type Output = MaskedCellBuffer;
        fn neg(self) -> Self::Output {
            Self::Output::new(self.buffer().neg(), self.mask().clone())
        }
    }
    impl Neg for MaskedCellBuffer {
        type Output = MaskedCellBuffer;
        fn neg(self) -> Self::Output {
            let (buf, mask) = self.into();
            Self::Output::new((&buf).neg(), mask)
        }
    }
}
