//! `Mask` (src/masked/mask.rs): the image of `Vec<bool>` in HBM, one byte per cell, 0 or 1.
use crate::ffi::*;
use crate::{must, stream, DeviceMem};
use std::ops::{BitAnd, BitOr, Not};
use std::os::raw::c_void;

pub struct Mask {
    pub(crate) len: usize,
    pub(crate) mem: DeviceMem,
}

impl Mask {
    pub(crate) fn uninit(len: usize) -> Self { Self { len, mem: DeviceMem::new(len) } }
    pub(crate) fn dev_ptr(&self) -> *const u8 { self.mem.ptr() as *const u8 }
    pub(crate) fn dev_ptr_mut(&self) -> *mut u8 { self.mem.ptr() as *mut u8 }

    /// src/masked/mask.rs:16-18
    pub fn new(values: Vec<bool>) -> Self {
        let out = Self::uninit(values.len());
        if !values.is_empty() {
            // `bool` is one byte holding 0 or 1: the Vec is already the device image
            must(unsafe { ec_upload(out.mem.ptr(), values.as_ptr() as *const c_void, values.len(), stream()) }, "ec_upload");
        }
        out
    }
    /// src/masked/mask.rs:21-23
    pub fn fill(len: usize, value: bool) -> Self {
        let out = Self::uninit(len);
        let v = crate::CellValue::UInt8(value as u8).to_ffi();
        must(unsafe { ec_fill(crate::CellType::UInt8 as u8, out.mem.ptr(), len, &v, stream()) }, "ec_fill");
        out
    }
    pub fn len(&self) -> usize { self.len }
    pub fn is_empty(&self) -> bool { self.len == 0 }

    /// src/masked/mask.rs:57-59
    pub fn get(&self, index: usize) -> bool {
        assert!(index < self.len, "index out of bounds: the len is {} but the index is {}", self.len, index);
        let mut b = 0u8;
        must(unsafe { ec_download(&mut b as *mut u8 as *mut c_void, self.dev_ptr().add(index) as *const c_void, 1, stream()) }, "ec_download");
        b != 0
    }
    /// (number of `true` cells, number of `false` cells) — src/masked/mask.rs:72-80
    pub fn counts(&self) -> (usize, usize) {
        let (mut t, mut f) = (0u64, 0u64);
        must(unsafe { ec_mask_counts(self.dev_ptr(), self.len, &mut t, &mut f, stream()) }, "ec_mask_counts");
        (t as usize, f as usize)
    }
    /// src/masked/mask.rs:67-69
    pub fn all(&self, value: bool) -> bool {
        let (t, f) = self.counts();
        if value { f == 0 } else { t == 0 }
    }
    pub fn to_vec(&self) -> Vec<bool> {
        let mut bytes = vec![0u8; self.len];
        if self.len > 0 {
            must(unsafe { ec_download(bytes.as_mut_ptr() as *mut c_void, self.dev_ptr() as *const c_void, self.len, stream()) }, "ec_download");
        }
        bytes.into_iter().map(|b| b != 0).collect()
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

// Borrowed forms zip (result length = the shorter operand); owned forms update the left operand in
// place and keep its length (src/masked/mask.rs:98-163).
impl BitAnd for &Mask {
    type Output = Mask;
    fn bitand(self, rhs: Self) -> Mask {
        let out = Mask::uninit(self.len.min(rhs.len));
        must(unsafe { ec_mask_and(self.dev_ptr(), rhs.dev_ptr(), out.len, out.dev_ptr_mut(), stream()) }, "ec_mask_and");
        out
    }
}
impl BitOr for &Mask {
    type Output = Mask;
    fn bitor(self, rhs: Self) -> Mask {
        let out = Mask::uninit(self.len.min(rhs.len));
        must(unsafe { ec_mask_or(self.dev_ptr(), rhs.dev_ptr(), out.len, out.dev_ptr_mut(), stream()) }, "ec_mask_or");
        out
    }
}
impl BitAnd for Mask {
    type Output = Mask;
    fn bitand(self, rhs: Self) -> Mask {
        let n = self.len.min(rhs.len);
        must(unsafe { ec_mask_and(self.dev_ptr(), rhs.dev_ptr(), n, self.dev_ptr_mut(), stream()) }, "ec_mask_and");
        self
    }
}
impl BitOr for Mask {
    type Output = Mask;
    fn bitor(self, rhs: Self) -> Mask {
        let n = self.len.min(rhs.len);
        must(unsafe { ec_mask_or(self.dev_ptr(), rhs.dev_ptr(), n, self.dev_ptr_mut(), stream()) }, "ec_mask_or");
        self
    }
}
impl Not for &Mask {
    type Output = Mask;
    fn not(self) -> Mask {
        let out = Mask::uninit(self.len);
        must(unsafe { ec_mask_not(self.dev_ptr(), self.len, out.dev_ptr_mut(), stream()) }, "ec_mask_not");
        out
    }
}
impl Not for Mask {
    type Output = Mask;
    fn not(self) -> Mask {
        must(unsafe { ec_mask_not(self.dev_ptr(), self.len, self.dev_ptr_mut(), stream()) }, "ec_mask_not");
        self
    }
}
