// source.hpp -- header-only C++ mirror of the reference's file-backed `Source` (quartiq/stabilizer-stream
// src/source.rs) over the C ABI, beside psd_cascade.hpp: same names, same granularity for get(), plus the batched
// feed() a GPU path needs.  UDP, the noise generator and the DSM source are host I/O outside the accelerated path
//  and are not mirrored.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/psdcascade.h"

namespace stabilizer_stream {

struct SourceOpts { // file-backed subset of SourceOpts (src/source.rs:15-48)
    std::optional<std::string> file;        // --file: frames file
    size_t frame_size = 8 + 30 * 2 * 6 * 4; // --frame-size default (src/source.rs:31)
    bool repeat = false;                    // --repeat
    std::optional<std::string> raw;         // --raw: single f32 trace, native endian
};

using Traces = std::vector<std::pair<const char *, std::vector<float>>>; // Vec<(&'static str, Vec<f32>)>

// Frame::from_bytes + Payload::traces on the host for the four formats (src/de/frame.rs:25-60, src/de/data.rs:11-212; format
// ids src/de/mod.rs:12-17); throws the de::Error texts (src/de/mod.rs:19-27).  Only get() uses it: bulk ingest decodes on the
// device (psdc_process_frames).  The f32 arithmetic is the reference's, operation by operation (the products and their sum kept
// apart: rustc never fuses them).
inline Traces decode_frame(const uint8_t *buf, size_t len, uint32_t *seq_out, uint32_t *batches_out, int *format_out = nullptr)
{
    if (len < 8)
        throw std::runtime_error("frame shorter than its header"); // &input[..HEADER_SIZE] panics (frame.rs:50)
    if (buf[0] != 0x7b || buf[1] != 0x05)
        throw std::runtime_error("Invalid frame header");
    const int fmt = buf[2];
    if (fmt < 1 || fmt > 4)
        throw std::runtime_error("Unknown format ID");
    const uint32_t batches = buf[3];
    const size_t bb = fmt == 1 ? 64 : fmt == 2 ? 56 : fmt == 3 ? 80 : 24; // bytes per batch (data.rs:13, 86, 144, 168)
    if ((len - 8) % bb != 0 || (len - 8) / bb != batches)
        throw std::runtime_error("Payload size");
    *seq_out = (uint32_t)buf[4] | ((uint32_t)buf[5] << 8) | ((uint32_t)buf[6] << 16) | ((uint32_t)buf[7] << 24);
    *batches_out = batches;
    if (format_out)
        *format_out = fmt;
    Traces out;
    if (fmt == 1) {
        const float lsb = 4.096f * 2.5f / 32768.0f; // src/de/data.rs:28-35
        static const char *names[4] = {"ADC0", "ADC1", "DAC0", "DAC1"};
        for (int c = 0; c < 4; ++c) {
            std::vector<float> v(8 * (size_t)batches);
            for (uint32_t b = 0; b < batches; ++b)
                for (int i = 0; i < 8; ++i) {
                    const uint8_t *p = buf + 8 + ((size_t)b * 4 + (size_t)c) * 16 + 2 * i; // [[[u8;2];8];4] per batch (data.rs:13)
                    uint16_t raw = (uint16_t)p[0] | ((uint16_t)p[1] << 8);
                    if (c >= 2)
                        raw = (uint16_t)(raw + 0x8000u); // i16.wrapping_add(i16::MIN) (data.rs:64,75)
                    v[8 * (size_t)b + i] = (float)(int16_t)raw * lsb;
                }
            out.emplace_back(names[c], std::move(v));
        }
        return out;
    }
    auto word = [&](uint32_t b, int i) {
        const uint8_t *q = buf + 8 + (size_t)b * bb + 4 * (size_t)i;
        return (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    };
    auto i32f = [&](uint32_t b, int i) { return (float)(int32_t)word(b, i); };
    auto hyp = [](float a, float c) { // (a as f32).powi(2) + (c as f32).powi(2), .sqrt()
        volatile float aa = a * a, cc = c * c;
        volatile float sum = aa + cc;
        return std::sqrt((float)sum);
    };
    const float two31 = 2147483648.0f, two32 = 4294967296.0f, tau = 6.28318530717958647692f;
    static const char *names[3][4] = {{"AR", "AP", "BI", "BQ"}, {"T00", "T20", "I0", "I1"},
                                      {"phase (rad)", "frequency (kHz)", "amplitude (V/G10)", nullptr}};
    const int ntr = fmt == 4 ? 3 : 4;
    std::vector<std::vector<float>> v(ntr, std::vector<float>(batches));
    for (uint32_t b = 0; b < batches; ++b) {
        if (fmt == 2) { // Fls::traces, data.rs:97-139
            v[0][b] = hyp(i32f(b, 0), i32f(b, 1)) * (1.0f / two31);
            const int64_t ph = (int64_t)((uint64_t)word(b, 2) | ((uint64_t)word(b, 3) << 32));
            v[1][b] = (float)ph * (tau / 65536.0f);
            v[2][b] = i32f(b, 7) / two31;
            v[3][b] = i32f(b, 8) / two31;
        } else if (fmt == 3) { // ThermostatEem::traces, data.rs:154-163
            static const int idx[4] = {0, 8, 13, 16};
            for (int t = 0; t < 4; ++t) {
                const uint32_t u = word(b, idx[t]);
                std::memcpy(&v[t][b], &u, 4);
            }
        } else { // Mpll::traces, data.rs:178-211
            v[0][b] = i32f(b, 4) * (tau / two32);
            v[1][b] = i32f(b, 5) * (1.0f / 1.28e-3f / two32);
            v[2][b] = hyp(i32f(b, 0), i32f(b, 1)) * (10.24f / 10.0f * 2.0f * 2.0f / two32);
        }
    }
    for (int t = 0; t < ntr; ++t)
        out.emplace_back(names[fmt - 2][t], std::move(v[t]));
    return out;
}
inline Traces decode_adcdac_frame(const uint8_t *buf, size_t len, uint32_t *seq_out, uint32_t *batches_out)
{
    if (len >= 8 && buf[0] == 0x7b && buf[1] == 0x05 && buf[2] >= 2 && buf[2] <= 4)
        throw std::runtime_error("Unknown format ID"); // (a valid id that is not AdcDac)
    return decode_frame(buf, len, seq_out, batches_out);
}

class Source { // Source::new / get / finish (src/source.rs:66-171) for Data::File and Data::Raw
public:
    explicit Source(SourceOpts opts) : opts_(std::move(opts))
    {
        if (opts_.file.has_value() == opts_.raw.has_value())
            throw std::invalid_argument("exactly one of file / raw (UDP, noise and dsm sources are out of scope)");
        f_ = std::fopen((opts_.file ? *opts_.file : *opts_.raw).c_str(), "rb");
        if (!f_)
            throw std::runtime_error("cannot open the source file");
    }
    Source(const Source &) = delete;
    Source &operator=(const Source &) = delete;
    ~Source()
    {
        if (f_)
            std::fclose(f_);
    }

    // One reference-sized chunk: Data::Raw at most 2048 bytes = 512 samples (src/source.rs:150-157), Data::File one
    // frame (:135-147); --repeat wraps at the end of the file.  End of file without --repeat, as in the reference:
    // Data::Raw keeps returning Ok with an EMPTY "raw" trace (`read` gives 0 bytes, :151-157 -- also for a tail
    // shorter than one f32); Data::File fails (`read_exact` -> UnexpectedEof, :137-146): get() returns false.
    // at_eof() tells a raw reader that the file is exhausted.
    bool at_eof() const { return eof_; }
    bool get(Traces &out)
    {
        uint8_t buf[2048];
        if (opts_.raw) {
            size_t len = std::fread(buf, 1, sizeof buf, f_);
            if (len == 0 && opts_.repeat) { // :152-155 (the reference spins on an empty file; one retry here)
                std::fseek(f_, 0, SEEK_SET);
                len = std::fread(buf, 1, sizeof buf, f_);
            }
            eof_ = len == 0;
            std::vector<float> v(len / 4); // cast_slice(&buf[..len / 4 * 4]) (:156)
            std::memcpy(v.data(), buf, v.size() * 4);
            out.clear();
            out.emplace_back("raw", std::move(v));
            return true;
        }
        if (opts_.frame_size > sizeof buf)
            throw std::runtime_error("frame_size exceeds the 2048-byte buffer (src/source.rs:136)");
        for (int pass = 0; pass < 2; ++pass) {
            const size_t len = std::fread(buf, 1, opts_.frame_size, f_);
            if (len < opts_.frame_size) { // read_exact: UnexpectedEof (:137-146)
                if (opts_.repeat && pass == 0) {
                    std::fseek(f_, 0, SEEK_SET);
                    continue;
                }
                return false;
            }
            uint32_t seq = 0, batches = 0;
            out = decode_frame(buf, opts_.frame_size, &seq, &batches);
            received_ += batches; // Loss::update (src/loss.rs:11-26)
            if (have_seq_)
                dropped_ += (uint32_t)(seq - next_seq_);
            next_seq_ = seq + batches;
            have_seq_ = true;
            return true;
        }
        return false;
    }

    // Batched path: read up to max_bytes of the SAME byte formats and ingest them in one ABI call (raw -> channel
    // `channel`, frames of any of the four formats -> channel i for trace i).  Returns the bytes consumed, 0 at EOF.
    size_t feed(psdc_handle *h, size_t max_bytes = (size_t)64 << 20, uint32_t channel = 0)
    {
        const size_t unit = opts_.raw ? 4 : opts_.frame_size;
        std::vector<uint8_t> buf(std::max<size_t>(1, max_bytes / unit) * unit);
        for (int pass = 0; pass < 2; ++pass) {
            const size_t len = std::fread(buf.data(), 1, buf.size(), f_) / unit * unit;
            if (len == 0) {
                if (opts_.repeat && pass == 0) {
                    std::fseek(f_, 0, SEEK_SET);
                    continue;
                }
                return 0;
            }
            int rc;
            if (opts_.raw) {
                rc = psdc_process(h, channel, reinterpret_cast<const float *>(buf.data()), len / 4);
            } else {
                size_t ok = 0;
                rc = psdc_process_frames(h, buf.data(), opts_.frame_size, len / opts_.frame_size, &ok);
            }
            if (rc < 0)
                throw std::runtime_error(std::string("psdcascade: ") + psdc_last_error(h));
            return len;
        }
        return 0;
    }

    // Loss::analyze (src/loss.rs:28-38): fraction of dropped batches on the get() path
    double finish() const { return received_ ? (double)dropped_ / (double)(received_ + dropped_) : 0.0; }
    uint64_t received() const { return received_; }
    uint64_t dropped() const { return dropped_; }

private:
    SourceOpts opts_;
    std::FILE *f_ = nullptr;
    uint64_t received_ = 0, dropped_ = 0;
    uint32_t next_seq_ = 0;
    bool have_seq_ = false;
    bool eof_ = false;
};

} // namespace stabilizer_stream
