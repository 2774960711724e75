//! `MaskedCellBuffer` (src/masked/masked_buffer.rs of the reference): a [`CellBuffer`] with a companion [`Mask`],
//! both device-resident.  Arithmetic computes the value op over ALL cells (masked-out ones included, as the reference
//! does — they take part in the derived `PartialEq`) and ANDs the masks, in one launch; `min_max`, `counts` and
//! `to_vec_with_nodata` are the places where the mask decides.
use crate::device::stream;
use crate::error::{check, must};
use crate::ffi::*;
use crate::{BufferOps, CellBuffer, CellEncoding, CellType, CellValue, Mask, NoData};
use std::fmt::{Debug, Formatter};

/// A [`CellBuffer`] with a companion [`Mask`] of the same length: `true` marks a valid cell, `false` a no-data cell.
#[derive(Clone, PartialEq, PartialOrd)]
pub struct MaskedCellBuffer(CellBuffer, Mask);

impl MaskedCellBuffer {
    /// Pair a buffer with its mask.
    ///
    /// # Panics
    /// When the two lengths differ.
    pub fn new(buffer: CellBuffer, mask: Mask) -> Self {
        assert_eq!(buffer.len(), mask.len(), "Mask and buffer must have the same length.");
        Self(buffer, mask)
    }

    /// `buffer` with every cell valid (what `from_vec`, `fill`, `with_defaults`, `fill_via` and `From<CellBuffer>` build).
    fn all_valid(buffer: CellBuffer) -> Self {
        let mask = Mask::fill(buffer.len(), true);
        Self(buffer, mask)
    }

    /// Upload `data` and derive the mask from a sentinel: `false` exactly where a cell equals `nodata` under the total
    /// order (one kernel over the uploaded cells; `IsNodata::is` does the same test for a single value on the host).
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

    /// The cell at `index` if it is valid, `None` if it is masked out (two one-cell downloads).
    pub fn get_masked(&self, index: usize) -> Option<CellValue> {
        self.1.get(index).then(|| self.0.get(index))
    }

    /// The cell at `index` together with its validity flag.
    pub fn get_with_mask(&self, index: usize) -> (CellValue, bool) {
        (self.0.get(index), self.1.get(index))
    }

    /// Overwrite cell and validity flag at `index`; `Err(NarrowingError)` (nothing written) when `value`'s cell type
    /// does not fit the buffer's.
    pub fn put_with_mask(&mut self, index: usize, value: CellValue, mask: bool) -> crate::error::Result<()> {
        self.0.put(index, value)?;
        self.1.put(index, mask);
        Ok(())
    }

    /// `(data, nodata)`: the numbers of valid and of masked-out cells (a device reduction).
    pub fn counts(&self) -> (usize, usize) {
        self.1.counts()
    }

    /// Download as `Vec<T>` with every masked-out cell replaced by `no_data.value()` (`NoData::None`: the cells as they are)
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
    // the four constructors: the cells as CellBuffer builds them, every one valid
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        Self::all_valid(CellBuffer::from_vec(data))
    }

    fn with_defaults(len: usize, ct: CellType) -> Self {
        Self::all_valid(CellBuffer::with_defaults(len, ct))
    }

    fn fill(len: usize, value: CellValue) -> Self {
        Self::all_valid(CellBuffer::fill(len, value))
    }

    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T,
    {
        Self::all_valid(CellBuffer::fill_via(len, f))
    }

    // the accessors look at the cells only
    fn len(&self) -> usize {
        self.0.len
    }

    fn cell_type(&self) -> CellType {
        self.0.ct
    }

    fn get(&self, index: usize) -> CellValue {
        self.0.get(index)
    }

    fn put(&mut self, idx: usize, value: CellValue) -> crate::error::Result<()> {
        self.0.put(idx, value)
    }

    /// The cells widened to `cell_type` (refused up front if that would narrow), the mask carried over.
    fn convert(&self, cell_type: CellType) -> crate::error::Result<Self>
    where
        Self: Sized,
    {
        self.0.convert(cell_type).map(|cells| Self(cells, self.1.clone()))
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

/// A plain buffer becomes a masked one with every cell valid.
impl From<CellBuffer> for MaskedCellBuffer {
    fn from(value: CellBuffer) -> Self {
        Self::all_valid(value)
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
