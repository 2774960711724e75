// raster_io.hpp — raster ingest for the host mirror: the step *before* the hot path.
//
// Mirrors the reference's GDAL adaptor (`RasterBandEx::{read_cells, read_cells_masked}`,
// src/gdal/rasterband.rs:82-125; type subset src/gdal/mod.rs:14-44; nodata f64 -> NoData<T>
// src/gdal/mod.rs:49-70).  libgdal is not available here, so the reader covers exactly what the
// reference's fixtures (testkit/data/*.tiff) need: classic little-endian TIFF, one sample per pixel,
// uncompressed, strip-organised, 8/16/32/64-bit unsigned / signed / IEEE cells, GDAL_NODATA in ASCII
// tag 42113.  Anything else is rejected with UnsupportedCellTypeError / Error.
//
// Cells go host -> HBM once (`CellBuffer::from_vec`); `read_cells_rows` reads one row-block so each
// rank of a sharded job uploads only its own shard.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <optional>
#include <string>
#include <vector>

#include "erased_cells.hpp"

namespace erased_cells {

struct UnsupportedCellTypeError : Error {  // Error::UnsupportedCellTypeError (src/error.rs:16-17)
    explicit UnsupportedCellTypeError(const std::string& what) : Error(EC_ERR_UNSUPPORTED_TYPE, "Unsupported cell-type " + what) {}
};
struct NoDataConversionError : Error {  // Error::NoDataConversionError (src/error.rs:22-23)
    NoDataConversionError(double nd, const char* ty)
        : Error(EC_ERR_ARG, "Unable to convert " + std::to_string(nd) + " into NoData<" + ty + ">::Value") {}
};

// num-traits `f64::to_<int>()`: Some(truncated) iff the value lies in the open range (MIN-1, MAX+1).
template <typename T>
inline std::optional<T> f64_to(double v) {
    if constexpr (std::is_floating_point<T>::value) {
        return static_cast<T>(v);  // float -> float is a plain `as`
    } else {
        if (std::isnan(v)) return std::nullopt;
        const double lo = static_cast<double>(std::numeric_limits<T>::lowest()) - 1.0;
        const double hi = static_cast<double>(std::numeric_limits<T>::max()) + 1.0;
        if (!(v > lo && v < hi)) return std::nullopt;
        return static_cast<T>(v);
    }
}

// TryFrom<GdalND> for NoData<T> (src/gdal/mod.rs:49-70)
template <typename T>
inline NoData<T> nodata_from_f64(const std::optional<double>& nd, const char* type_name) {
    if (!nd) return NoData<T>::None();
    auto v = f64_to<T>(*nd);
    if (!v) throw NoDataConversionError(*nd, type_name);
    return NoData<T>::new_(*v);
}

class RasterBand {
    size_t width_ = 0, height_ = 0;
    CellType ct_ = CellType::UInt8;
    std::optional<double> no_data_;
    std::vector<uint8_t> cells_;  // row-major, host

    static uint16_t rd16(const std::vector<uint8_t>& d, size_t o) { uint16_t v; std::memcpy(&v, d.data() + o, 2); return v; }
    static uint32_t rd32(const std::vector<uint8_t>& d, size_t o) { uint32_t v; std::memcpy(&v, d.data() + o, 4); return v; }

public:
    // Dataset::open(path)?.rasterband(1)
    static RasterBand open(const std::string& path) {
        std::ifstream f(path, std::ios::binary);
        if (!f) throw Error(EC_ERR_ARG, "cannot open " + path);
        std::vector<uint8_t> d((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (d.size() < 8 || d[0] != 'I' || d[1] != 'I' || rd16(d, 2) != 42) throw Error(EC_ERR_ARG, path + ": not a little-endian classic TIFF");
        const size_t ifd = rd32(d, 4);
        const size_t nent = rd16(d, ifd);
        static const size_t tsz[17] = {0, 1, 1, 2, 4, 8, 1, 1, 2, 4, 8, 4, 8, 0, 0, 0, 8};
        std::map<int, std::vector<uint64_t>> tags;
        std::string nodata_txt;
        for (size_t i = 0; i < nent; ++i) {
            const size_t e = ifd + 2 + 12 * i;
            const int tag = rd16(d, e), typ = rd16(d, e + 2);
            const size_t cnt = rd32(d, e + 4);
            if (typ <= 0 || typ > 16 || tsz[typ] == 0) continue;
            const size_t bytes = tsz[typ] * cnt;
            const size_t pos = bytes > 4 ? rd32(d, e + 8) : e + 8;
            if (pos + bytes > d.size()) throw Error(EC_ERR_ARG, path + ": truncated TIFF");
            if (tag == 42113 && typ == 2) { nodata_txt.assign(reinterpret_cast<const char*>(d.data() + pos), cnt); continue; }
            std::vector<uint64_t> vals;
            for (size_t k = 0; k < cnt && (typ == 3 || typ == 4 || typ == 1); ++k)
                vals.push_back(typ == 3 ? rd16(d, pos + 2 * k) : typ == 4 ? rd32(d, pos + 4 * k) : d[pos + k]);
            tags[tag] = vals;
        }
        auto one = [&](int tag, uint64_t dflt) { auto it = tags.find(tag); return it == tags.end() || it->second.empty() ? dflt : it->second[0]; };
        RasterBand b;
        b.width_ = one(256, 0);
        b.height_ = one(257, 0);
        const uint64_t bits = one(258, 1), comp = one(259, 1), spp = one(277, 1), fmt = one(339, 1);
        if (comp != 1 || spp != 1) throw Error(EC_ERR_ARG, path + ": only uncompressed single-band TIFFs are supported");
        // TryFrom<GdalDataType> for CellType (src/gdal/mod.rs:30-44): the 7 types older GDALs know
        if (fmt == 1 && bits == 8) b.ct_ = CellType::UInt8;
        else if (fmt == 1 && bits == 16) b.ct_ = CellType::UInt16;
        else if (fmt == 1 && bits == 32) b.ct_ = CellType::UInt32;
        else if (fmt == 2 && bits == 16) b.ct_ = CellType::Int16;
        else if (fmt == 2 && bits == 32) b.ct_ = CellType::Int32;
        else if (fmt == 3 && bits == 32) b.ct_ = CellType::Float32;
        else if (fmt == 3 && bits == 64) b.ct_ = CellType::Float64;
        else throw UnsupportedCellTypeError("sample format " + std::to_string(fmt) + " with " + std::to_string(bits) + " bits");
        const auto& offs = tags[273];
        const auto& cnts = tags[279];
        if (offs.empty() || offs.size() != cnts.size()) throw Error(EC_ERR_ARG, path + ": missing strip tables");
        b.cells_.reserve(b.width_ * b.height_ * size_of(b.ct_));
        for (size_t s = 0; s < offs.size(); ++s) {
            if (offs[s] + cnts[s] > d.size()) throw Error(EC_ERR_ARG, path + ": strip outside the file");
            b.cells_.insert(b.cells_.end(), d.begin() + offs[s], d.begin() + offs[s] + cnts[s]);
        }
        if (b.cells_.size() != b.width_ * b.height_ * size_of(b.ct_)) throw Error(EC_ERR_ARG, path + ": strips do not cover the raster");
        if (!nodata_txt.empty()) b.no_data_ = std::strtod(nodata_txt.c_str(), nullptr);
        return b;
    }

    std::pair<size_t, size_t> size() const { return {width_, height_}; }  // raster_size()
    CellType band_type() const { return ct_; }
    std::optional<double> no_data_value() const { return no_data_; }

    // read_cells for rows [row0, row0 + nrows) — the whole band by default (src/gdal/rasterband.rs:82-103).
    CellBuffer read_cells_rows(size_t row0, size_t nrows) const {
        if (row0 + nrows > height_) throw std::out_of_range("row window outside the raster");
        const size_t sz = size_of(ct_), n = nrows * width_;
        const uint8_t* p = cells_.data() + row0 * width_ * sz;
        switch (ct_) {
#define EC_RD(ID, P) case CellType::ID: { std::vector<P> v(n); std::memcpy(v.data(), p, n * sz); return CellBuffer::from_vec(v); }
            EC_HOST_WITH_CT(EC_RD)
#undef EC_RD
        }
        throw UnsupportedCellTypeError(to_string(ct_));
    }
    CellBuffer read_cells() const { return read_cells_rows(0, height_); }

    // read_cells_masked (src/gdal/rasterband.rs:104-125): mask from the band's nodata value.
    MaskedCellBuffer read_cells_masked_rows(size_t row0, size_t nrows) const {
        if (row0 + nrows > height_) throw std::out_of_range("row window outside the raster");
        const size_t sz = size_of(ct_), n = nrows * width_;
        const uint8_t* p = cells_.data() + row0 * width_ * sz;
        switch (ct_) {
#define EC_RDM(ID, P) case CellType::ID: { std::vector<P> v(n); std::memcpy(v.data(), p, n * sz); \
            return MaskedCellBuffer::from_vec_with_nodata(v, nodata_from_f64<P>(no_data_, #P)); }
            EC_HOST_WITH_CT(EC_RDM)
#undef EC_RDM
        }
        throw UnsupportedCellTypeError(to_string(ct_));
    }
    MaskedCellBuffer read_cells_masked() const { return read_cells_masked_rows(0, height_); }
};

}  // namespace erased_cells
