//! [`Mask`]: the image of a `Vec<bool>` in HBM, one byte per cell, 0 or 1.
//!
//! PROVENANCE.  The type's name, its method signatures and its trait / operator impl headers are the reference's public
//! surface (erased-cells 0.1.1, src/masked/mask.rs:10-178, MIT License, Copyright (c) 2023 Astraea, Inc.); the bodies
//! and the host-shadow mechanism are this crate's.  See INTEGRATION.md §2.
//!
//! The reference hands out `&mut bool` (`iter_mut`, `IndexMut`): those need host memory.  A `Mask` therefore keeps
//! an optional host shadow: `iter_mut` / `index_mut` fill it from the device, mark it dirty and lend references
//! into it; every device-side use (`dev_ptr`) first writes a dirty shadow back.  Masks that are only ever built
//! and combined by kernels never materialise the shadow.
use crate::device::{download, stream, upload, DeviceMem};
use crate::error::must;
use crate::ffi::*;
use crate::Elided;
use std::cell::Cell;
use std::cmp::Ordering;
use std::fmt::{Debug, Formatter};
use std::ops::{BitAnd, BitOr, Index, IndexMut, Not};
use std::os::raw::c_void;
use std::vec::IntoIter;

/// Which cells of a [`MaskedCellBuffer`][super::MaskedCellBuffer] are data (`true`) and which are no-data (`false`).
pub struct Mask {
    pub(crate) len: usize,
    pub(crate) mem: DeviceMem,
    shadow: Vec<bool>,  // host copy lent out by iter_mut / index_mut; empty = not materialised
    dirty: Cell<bool>,  // the shadow has been lent mutably since it was last written back
}

impl Mask {
    pub(crate) fn uninit(len: usize) -> Self {
        Self { len, mem: DeviceMem::new(len), shadow: Vec::new(), dirty: Cell::new(false) }
    }
    /// Device pointer for reading; a dirty host shadow is written back first.
    pub(crate) fn dev_ptr(&self) -> *const u8 {
        if self.dirty.get() {
            upload(self.mem.ptr(), &self.shadow);
            self.dirty.set(false);
        }
        self.mem.ptr() as *const u8
    }
    /// Device pointer a kernel is about to write through (owned, in-place forms): the shadow no longer describes it.
    fn dev_ptr_overwritten(&mut self) -> *mut u8 {
        let p = self.dev_ptr() as *mut u8;
        self.shadow.clear();
        p
    }
    pub(crate) fn dev_ptr_mut(&self) -> *mut u8 {
        self.mem.ptr() as *mut u8
    }
    fn load_shadow(&mut self) {
        if self.shadow.len() != self.len {
            self.shadow = self.to_vec();
        }
    }
    /// The mask as host values (one download).
    pub(crate) fn to_vec(&self) -> Vec<bool> {
        download::<u8>(self.dev_ptr() as *const c_void, self.len).into_iter().map(|b| b != 0).collect()
    }

    /// Upload `values` as they are.
    pub fn new(values: Vec<bool>) -> Self {
        let out = Self::uninit(values.len());
        upload(out.mem.ptr(), &values); // `bool` is one byte holding 0 or 1: the Vec is already the device image
        out
    }

    /// `len` flags, all `value` (a fill kernel; nothing is uploaded).
    pub fn fill(len: usize, value: bool) -> Self {
        let out = Self::uninit(len);
        let v = crate::CellValue::UInt8(value as u8).to_ffi();
        must(unsafe { ec_fill(crate::CellType::UInt8 as u8, out.mem.ptr(), len, &v, stream()) }, "ec_fill");
        out
    }

    /// Flag `i` is `f(i)`: evaluated on the host, uploaded once.
    pub fn fill_via<F>(len: usize, f: F) -> Self
    where
        F: Fn(usize) -> bool,
    {
        Self::new((0..len).map(f).collect())
    }

    /// Number of flags.
    pub fn len(&self) -> usize {
        self.len
    }

    /// `len() == 0`.
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }

    /// Overwrite one flag (a one-byte upload; the host shadow, if there is one, follows).
    ///
    /// # Panics
    /// If `index >= self.len()`.
    pub fn put(&mut self, index: usize, value: bool) {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
        let p = unsafe { (self.dev_ptr() as *mut u8).add(index) } as *mut c_void;
        upload(p, &[value]);
        if !self.shadow.is_empty() {
            self.shadow[index] = value;
        }
    }

    /// One flag: from the host shadow when it is current, else a one-byte download.
    ///
    /// # Panics
    /// If `index >= self.len()`.
    pub fn get(&self, index: usize) -> bool {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
        if self.shadow.len() == self.len {
            return self.shadow[index];
        }
        download::<u8>(unsafe { self.dev_ptr().add(index) } as *const c_void, 1)[0] != 0
    }

    /// `&mut bool` access needs host memory: the flags are brought into the host shadow, lent out from there, and
    /// written back before the device next reads the mask.
    pub fn iter_mut(&mut self) -> impl Iterator<Item = &'_ mut bool> {
        self.load_shadow();
        self.dirty.set(true);
        self.shadow.iter_mut()
    }

    /// Are all flags `value`?  (Answered from `counts`: one device reduction.)
    pub fn all(&self, value: bool) -> bool {
        let (data, nodata) = self.counts();
        if value { nodata == 0 } else { data == 0 }
    }

    /// `(data, nodata)` = (number of `true`, number of `false`) flags — a device reduction over the bytes.
    pub fn counts(&self) -> (usize, usize) {
        let (mut t, mut f) = (0u64, 0u64);
        must(unsafe { ec_mask_counts(self.dev_ptr(), self.len, &mut t, &mut f, stream()) }, "ec_mask_counts");
        (t as usize, f as usize)
    }
}

impl Default for Mask {
    fn default() -> Self {
        Self::uninit(0)
    }
}

impl Clone for Mask {
    fn clone(&self) -> Self {
        let out = Self::uninit(self.len);
        if self.len > 0 {
            must(unsafe { ec_copy(out.mem.ptr(), self.dev_ptr() as *const c_void, self.len, stream()) }, "ec_copy");
        }
        out
    }
}

// derived `PartialEq, PartialOrd, Ord, Eq` of `Mask(Vec<bool>)` (src/masked/mask.rs:10): lexicographic over the
// bytes, then length — decided on the device like `CellBuffer`'s ordering.
impl Ord for Mask {
    fn cmp(&self, other: &Self) -> Ordering {
        let mut o = 0i32;
        let u8t = crate::CellType::UInt8 as u8;
        must(
            unsafe {
                ec_buffer_cmp(u8t, self.dev_ptr() as *const c_void, self.len, u8t, other.dev_ptr() as *const c_void, other.len, &mut o, stream())
            },
            "ec_buffer_cmp",
        );
        o.cmp(&0)
    }
}
impl PartialOrd for Mask {
    fn partial_cmp(&self, other: &Self) -> Option<Ordering> {
        Some(self.cmp(other))
    }
}
impl PartialEq for Mask {
    fn eq(&self, other: &Self) -> bool {
        self.cmp(other) == Ordering::Equal
    }
}
impl Eq for Mask {}

impl Extend<bool> for Mask {
    fn extend<T: IntoIterator<Item = bool>>(&mut self, iter: T) {
        let tail: Vec<bool> = iter.into_iter().collect();
        let grown = Mask::uninit(self.len + tail.len());
        if self.len > 0 {
            must(unsafe { ec_copy(grown.mem.ptr(), self.dev_ptr() as *const c_void, self.len, stream()) }, "ec_copy");
        }
        upload(unsafe { grown.dev_ptr_mut().add(self.len) } as *mut c_void, &tail);
        *self = grown;
    }
}

impl Index<usize> for Mask {
    type Output = bool;

    fn index(&self, index: usize) -> &Self::Output {
        if self.get(index) { &true } else { &false }
    }
}

impl IndexMut<usize> for Mask {
    fn index_mut(&mut self, index: usize) -> &mut Self::Output {
        self.load_shadow();
        self.dirty.set(true);
        &mut self.shadow[index]
    }
}

// Borrowed forms zip (result length = the shorter operand); owned forms update the left operand in place and
// keep its length (src/masked/mask.rs:103-163).
impl Not for Mask {
    type Output = Mask;
    fn not(mut self) -> Self::Output {
        let p = self.dev_ptr_overwritten();
        must(unsafe { ec_mask_not(p, self.len, p, stream()) }, "ec_mask_not");
        self
    }
}

impl Not for &Mask {
    type Output = Mask;
    fn not(self) -> Self::Output {
        let out = Mask::uninit(self.len);
        must(unsafe { ec_mask_not(self.dev_ptr(), self.len, out.dev_ptr_mut(), stream()) }, "ec_mask_not");
        out
    }
}

impl BitAnd for Mask {
    type Output = Self;
    fn bitand(mut self, rhs: Self) -> Self::Output {
        let n = self.len.min(rhs.len);
        let p = self.dev_ptr_overwritten();
        must(unsafe { ec_mask_and(p, rhs.dev_ptr(), n, p, stream()) }, "ec_mask_and");
        self
    }
}

impl BitAnd for &Mask {
    type Output = Mask;
    fn bitand(self, rhs: Self) -> Self::Output {
        let out = Mask::uninit(self.len.min(rhs.len));
        must(unsafe { ec_mask_and(self.dev_ptr(), rhs.dev_ptr(), out.len, out.dev_ptr_mut(), stream()) }, "ec_mask_and");
        out
    }
}

impl BitOr for Mask {
    type Output = Self;
    fn bitor(mut self, rhs: Self) -> Self::Output {
        let n = self.len.min(rhs.len);
        let p = self.dev_ptr_overwritten();
        must(unsafe { ec_mask_or(p, rhs.dev_ptr(), n, p, stream()) }, "ec_mask_or");
        self
    }
}

impl BitOr for &Mask {
    type Output = Mask;
    fn bitor(self, rhs: Self) -> Self::Output {
        let out = Mask::uninit(self.len.min(rhs.len));
        must(unsafe { ec_mask_or(self.dev_ptr(), rhs.dev_ptr(), out.len, out.dev_ptr_mut(), stream()) }, "ec_mask_or");
        out
    }
}

impl Debug for Mask {
    fn fmt(&self, f: &mut Formatter<'_>) -> std::fmt::Result {
        if self.len > 10 {
            let all = |lo: usize, n: usize| -> Vec<bool> {
                download::<u8>(unsafe { self.dev_ptr().add(lo) } as *const c_void, n).into_iter().map(|b| b != 0).collect()
            };
            let (head, tail) = (all(0, 5), all(self.len - 5, 5));
            f.write_fmt(format_args!("Mask({:?}, ... {:?})", Elided(&head), Elided(&tail)))
        } else {
            f.write_fmt(format_args!("Mask({:?})", Elided(&self.to_vec())))
        }
    }
}

impl IntoIterator for Mask {
    type Item = bool;
    type IntoIter = IntoIter<bool>;
    fn into_iter(self) -> Self::IntoIter {
        self.to_vec().into_iter()
    }
}
