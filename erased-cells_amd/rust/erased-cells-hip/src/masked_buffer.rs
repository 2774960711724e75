//! [`MaskedCellBuffer`]: cells and their validity mask, both resident in HBM.
//!
//! PROVENANCE.  The type's name, its method signatures and its trait-impl headers are the reference's public surface
//! (erased-cells 0.1.1, src/masked/masked_buffer.rs:39-383, MIT License, Copyright (c) 2023 Astraea, Inc.) — kept so
//! that code written against the reference compiles unchanged; what the methods DO is written here against the C ABI.
//! See INTEGRATION.md §2 for the line ranges.
//!
//! Semantics carried over from the reference: arithmetic computes the value op over ALL cells —
//! masked-out ones included, they take part in the derived `PartialEq` — and ANDs the masks, here in ONE launch
//! (`ec_masked_binop`); `min_max`, `counts` and `to_vec_with_nodata` are the places where the mask decides.
use crate::device::stream;
use crate::error::{check, must, Result};
use crate::ffi::*;
use crate::{BufferOps, CellBuffer, CellBufferIterator, CellEncoding, CellType, CellValue, Mask, NoData};
use std::fmt::{Debug, Formatter};
use std::ops::{Add, Div, Mul, Neg, Sub};

/// A [`CellBuffer`] paired with a [`Mask`] of the same length: `true` = the cell is data, `false` = no-data.
#[derive(Clone, PartialEq, PartialOrd)]
pub struct MaskedCellBuffer(CellBuffer, Mask);

impl MaskedCellBuffer {
    /// # Panics
    /// If `buffer` and `mask` differ in length.
    pub fn new(buffer: CellBuffer, mask: Mask) -> Self {
        assert_eq!(buffer.len(), mask.len(), "Mask and buffer must have the same length.");
        MaskedCellBuffer(buffer, mask)
    }

    /// Wrap `cells` with a mask that is `true` everywhere (one `ec_fill` of `cells.len()` bytes).
    fn unmasked(cells: CellBuffer) -> Self {
        let everywhere = Mask::fill(cells.len(), true);
        MaskedCellBuffer(cells, everywhere)
    }

    /// Upload `data`; the mask is `false` exactly where a cell equals the marker of `nodata` under the total order.
    /// One kernel over the uploaded cells writes the mask bytes (`ec_mask_from_nodata`); with `NoData::None` the ABI
    /// receives a null marker and writes `true` everywhere.
    pub fn from_vec_with_nodata<T: CellEncoding>(data: Vec<T>, nodata: NoData<T>) -> Self {
        let cells = CellBuffer::from_vec(data);
        let mask = Mask::uninit(cells.len());
        let marker = nodata.to_ffi();
        let marker_ptr = match marker.as_ref() {
            Some(v) => v as *const ec_value,
            None => std::ptr::null(),
        };
        must(
            unsafe { ec_mask_from_nodata(cells.ct as u8, cells.dev_ptr(), cells.len(), marker_ptr, mask.dev_ptr_mut(), stream()) },
            "ec_mask_from_nodata",
        );
        MaskedCellBuffer(cells, mask)
    }

    /// `mv(i)` gives cell `i` and its validity; both vectors are built on the host and uploaded once each.
    pub fn fill_with_mask_via<T, F>(len: usize, mv: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> (T, bool),
    {
        let (cells, valid): (Vec<T>, Vec<bool>) = (0..len).map(mv).unzip();
        MaskedCellBuffer(CellBuffer::from_vec(cells), Mask::new(valid))
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

    /// `Some(cell)` where the mask says data, `None` where it says no-data.  The mask byte is fetched first; the cell
    /// is only downloaded when it is wanted.
    pub fn get_masked(&self, index: usize) -> Option<CellValue> {
        match self.1.get(index) {
            true => Some(self.0.get(index)),
            false => None,
        }
    }

    /// The cell at `index` and whether it is data.
    pub fn get_with_mask(&self, index: usize) -> (CellValue, bool) {
        let valid = self.1.get(index);
        (self.0.get(index), valid)
    }

    /// Store a cell and its validity.  A `value` that does not fit the buffer's cell type is refused before anything
    /// is written (`Err(NarrowingError)`).
    pub fn put_with_mask(&mut self, index: usize, value: CellValue, mask: bool) -> Result<()> {
        self.0.put(index, value).map(|()| self.1.put(index, mask))
    }

    /// `(data, nodata)` cell counts — a device reduction over the mask bytes.
    pub fn counts(&self) -> (usize, usize) {
        self.1.counts()
    }

    /// The cells as `Vec<T>` with the marker of `no_data` written over every masked-out cell: widen on the device
    /// (`ec_convert`, refused if `T` is narrower), select on the device (`ec_mask_select`), download once.
    /// `NoData::None` has no marker: the cells come back as they are.
    pub fn to_vec_with_nodata<T: CellEncoding>(self, no_data: NoData<T>) -> Result<Vec<T>> {
        let MaskedCellBuffer(cells, mask) = self;
        let widened = cells.convert(T::cell_type())?;
        let Some(marker) = no_data.to_ffi() else {
            return widened.to_vec::<T>();
        };
        if widened.is_empty() {
            return widened.to_vec::<T>(); // nothing to select; `to_vec` keeps the reference's behaviour for empty buffers
        }
        let picked = CellBuffer::uninit(T::cell_type(), widened.len());
        check(unsafe {
            ec_mask_select(T::cell_type() as u8, widened.dev_ptr(), mask.dev_ptr(), widened.len(), &marker, picked.mem.ptr(), stream())
        })?;
        picked.to_vec::<T>()
    }

    /// `self op rhs` cell by cell over the shorter length, masks ANDed — one `ec_masked_binop` launch.
    fn zip_op(&self, op: ec_op, rhs: &MaskedCellBuffer) -> MaskedCellBuffer {
        let n = usize::min(self.len(), rhs.len());
        if n == 0 {
            return MaskedCellBuffer(CellBuffer::empty_u8(), Mask::uninit(0)); // collecting nothing gives a UInt8 buffer
        }
        let values = CellBuffer::uninit(CellType::Float64, n);
        let valid = Mask::uninit(n);
        let (l, r) = (&self.0, &rhs.0);
        must(
            unsafe {
                ec_masked_binop(op, l.ct as u8, l.dev_ptr(), self.1.dev_ptr(), r.ct as u8, r.dev_ptr(), rhs.1.dev_ptr(), n,
                                values.mem.ptr() as *mut f64, valid.dev_ptr_mut(), stream())
            },
            "ec_masked_binop",
        );
        MaskedCellBuffer(values, valid)
    }

    /// `self op scalar`: the cells through `ec_binop_scalar`, the mask moved over untouched.
    fn scalar_op(self, op: ec_op, rhs: CellValue) -> MaskedCellBuffer {
        let MaskedCellBuffer(cells, mask) = self;
        MaskedCellBuffer::new(cells.binop_scalar(op, rhs), mask)
    }
}

impl BufferOps for MaskedCellBuffer {
    fn from_vec<T: CellEncoding>(data: Vec<T>) -> Self {
        Self::unmasked(data.into())
    }

    fn with_defaults(len: usize, ct: CellType) -> Self {
        Self::unmasked(CellBuffer::with_defaults(len, ct))
    }

    fn fill(len: usize, value: CellValue) -> Self {
        Self::unmasked(CellBuffer::fill(len, value))
    }

    fn fill_via<T, F>(len: usize, f: F) -> Self
    where
        T: CellEncoding,
        F: Fn(usize) -> T,
    {
        Self::unmasked(CellBuffer::fill_via(len, f))
    }

    fn len(&self) -> usize {
        self.0.len
    }

    fn cell_type(&self) -> CellType {
        self.0.ct
    }

    fn get(&self, index: usize) -> CellValue {
        self.0.get(index)
    }

    fn put(&mut self, idx: usize, value: CellValue) -> Result<()> {
        self.0.put(idx, value)
    }

    /// Widen the cells (refused up front when `cell_type` is narrower); the mask is copied device to device.
    fn convert(&self, cell_type: CellType) -> Result<Self>
    where
        Self: Sized,
    {
        let widened = self.0.convert(cell_type)?;
        Ok(MaskedCellBuffer(widened, self.1.clone()))
    }

    /// Extremes of the cells whose mask is `true`, from one masked device reduction.  With every cell masked out the
    /// fold never leaves its start values, so the pair comes back inverted: `(T::MAX, T::MIN)`.
    fn min_max(&self) -> (CellValue, CellValue) {
        let mut lo = CellValue::UInt8(0).to_ffi();
        let mut hi = lo;
        must(
            unsafe { ec_min_max(self.0.ct as u8, self.0.dev_ptr(), self.1.dev_ptr(), self.0.len, &mut lo, &mut hi, stream()) },
            "ec_min_max",
        );
        (CellValue::from_ffi(&lo), CellValue::from_ffi(&hi))
    }

    /// The mask plays no part here; see [`MaskedCellBuffer::to_vec_with_nodata`].
    fn to_vec<T: CellEncoding>(self) -> Result<Vec<T>> {
        self.0.to_vec()
    }
}

impl Debug for MaskedCellBuffer {
    /// `Float64MaskedCellBuffer(<cells>, <mask>)`, each part rendered by its own `Debug` (elided past ten items).
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        let name = format!("{}MaskedCellBuffer", self.0.ct);
        let mut tuple = f.debug_tuple(&name);
        tuple.field(&self.0);
        tuple.field(&self.1);
        tuple.finish()
    }
}

impl From<MaskedCellBuffer> for (CellBuffer, Mask) {
    fn from(value: MaskedCellBuffer) -> Self {
        let MaskedCellBuffer(cells, mask) = value;
        (cells, mask)
    }
}

impl<'a> From<&'a MaskedCellBuffer> for (&'a CellBuffer, &'a Mask) {
    fn from(value: &'a MaskedCellBuffer) -> Self {
        (value.buffer(), value.mask())
    }
}

impl From<CellBuffer> for MaskedCellBuffer {
    /// Every cell counts as data.
    fn from(value: CellBuffer) -> Self {
        MaskedCellBuffer::unmasked(value)
    }
}

impl<C: CellEncoding> FromIterator<C> for MaskedCellBuffer {
    fn from_iter<T: IntoIterator<Item = C>>(iter: T) -> Self {
        let cells: Vec<C> = Vec::from_iter(iter);
        MaskedCellBuffer::unmasked(cells.into())
    }
}

impl<C: CellEncoding> FromIterator<(C, bool)> for MaskedCellBuffer {
    /// The cell type is `C`'s even when the iterator is empty (both halves are typed vectors before they are uploaded).
    fn from_iter<T: IntoIterator<Item = (C, bool)>>(iter: T) -> Self {
        let mut cells: Vec<C> = Vec::new();
        let mut valid: Vec<bool> = Vec::new();
        for (c, m) in iter {
            cells.push(c);
            valid.push(m);
        }
        MaskedCellBuffer(cells.into(), Mask::new(valid))
    }
}

impl<C: CellEncoding> Extend<(C, bool)> for MaskedCellBuffer {
    /// Both halves grow by one device reallocation per call, however many items arrive.
    fn extend<T: IntoIterator<Item = (C, bool)>>(&mut self, iter: T) {
        let mut cells: Vec<C> = Vec::new();
        let mut valid: Vec<bool> = Vec::new();
        for (c, m) in iter {
            cells.push(c);
            valid.push(m);
        }
        self.0.extend(cells);
        self.1.extend(valid);
    }
}

impl<'buf> IntoIterator for &'buf MaskedCellBuffer {
    type Item = (CellValue, bool);
    type IntoIter = MaskedCellBufferIterator<'buf>;

    /// Two downloads (cells, mask bytes) serve the whole walk.
    fn into_iter(self) -> Self::IntoIter {
        MaskedCellBufferIterator { cells: (&self.0).into_iter(), valid: self.1.to_vec().into_iter() }
    }
}

/// Walks a [`MaskedCellBuffer`] as `(cell, is_data)` pairs over host copies of both halves.
pub struct MaskedCellBufferIterator<'buf> {
    cells: CellBufferIterator<'buf>,
    valid: std::vec::IntoIter<bool>,
}

impl Iterator for MaskedCellBufferIterator<'_> {
    type Item = (CellValue, bool);

    fn next(&mut self) -> Option<Self::Item> {
        let cell = self.cells.next()?;
        let is_data = self.valid.next()?;
        Some((cell, is_data))
    }
}

// api-surface(src/masked/masked_buffer.rs:323-383): the sixteen binary-operator impl headers and the two `Neg` impls
// (`&a op &b`, `a op b`, `a op &b`, `a op scalar` for + - * /; there is no `scalar op a` in the reference either)
macro_rules! masked_operator {
    ($trt:ident, $mth:ident, $code:expr) => {
        impl $trt for &MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: Self) -> Self::Output {
                self.zip_op($code, rhs)
            }
        }
        impl $trt for MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: Self) -> Self::Output {
                self.zip_op($code, &rhs)
            }
        }
        impl $trt<&MaskedCellBuffer> for MaskedCellBuffer {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: &MaskedCellBuffer) -> Self::Output {
                self.zip_op($code, rhs)
            }
        }
        impl<R> $trt<R> for MaskedCellBuffer
        where
            R: Into<CellValue>,
        {
            type Output = MaskedCellBuffer;
            fn $mth(self, rhs: R) -> Self::Output {
                self.scalar_op($code, rhs.into())
            }
        }
    };
}
masked_operator!(Add, add, EC_ADD);
masked_operator!(Sub, sub, EC_SUB);
masked_operator!(Mul, mul, EC_MUL);
masked_operator!(Div, div, EC_DIV);

impl Neg for &MaskedCellBuffer {
    type Output = MaskedCellBuffer;
    fn neg(self) -> Self::Output {
        MaskedCellBuffer(-&self.0, self.1.clone())
    }
}

impl Neg for MaskedCellBuffer {
    type Output = MaskedCellBuffer;
    fn neg(self) -> Self::Output {
        let MaskedCellBuffer(cells, mask) = self;
        MaskedCellBuffer(-&cells, mask)
    }
}
// end api-surface
