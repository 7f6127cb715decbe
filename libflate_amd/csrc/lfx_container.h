// lfx_container.h — gzip / zlib container header parse, ONE source for the device kernel (batch and one-shot
// decodes: one lane per stream) and for the host stream decoder (header-first construction, header getters).
//
// Reference behaviour reproduced (sile/libflate v2.3.0):
//   src/zlib.rs:221-266   Header::read_from: (CMF*256 + FLG) % 31, CM = 8, CINFO <= 7, FDICT rejected
//   src/gzip.rs:390-446   Header::read_from: ID1/ID2, CM = 8, FEXTRA / FNAME / FCOMMENT / FHCRC
//   src/gzip.rs:470-485   ExtraField::read_from (sub-fields must tile XLEN exactly)
//   src/gzip.rs:343-367   the HCRC covers the header RE-SERIALISED with only the five known flag bits, FLG.HCRC
//                         cleared and XFL passed through CompressionLevel::{from_u8,to_u8} (gzip.rs:69-82)
#pragma once
#include <stdint.h>

#include "lfx_common.h"
#include "lfx_decode.h"

namespace lfx {

// where the optional fields sit inside the header bytes (offsets relative to the stream start; len 0 = absent)
struct ContainerFields {
    uint64_t extra_off, extra_len;       // FEXTRA payload (after XLEN)
    uint64_t name_off, name_len;         // FNAME incl. the terminating NUL
    uint64_t comment_off, comment_len;   // FCOMMENT incl. the terminating NUL
    uint32_t mtime;
    uint8_t flg, xfl, os, cmf;           // gzip FLG / XFL / OS; zlib CMF (flg = zlib FLG)
};

// status 0: header ok, deflate_off = first DEFLATE byte.  status 2 (UnexpectedEof): more bytes might complete it.
LFX_HD inline DecHeader parse_container(int format, const uint8_t *p, uint64_t n, ContainerFields *f) {
    DecHeader h;
    h.status = 0; h.err = 0; h.a0 = 0; h.a1 = 0; h.deflate_off = 0; h.flags = 0; h._pad = 0;
    ContainerFields cf;
    cf.extra_off = cf.extra_len = cf.name_off = cf.name_len = cf.comment_off = cf.comment_len = 0;
    cf.mtime = 0; cf.flg = cf.xfl = cf.os = cf.cmf = 0;
    if (format == 1) {
        if (n < 2) { h.status = 2; h.err = ERR_EOF; h.deflate_off = n; }
        else {
            const uint32_t cmf = p[0], flg = p[1];
            cf.cmf = (uint8_t)cmf; cf.flg = (uint8_t)flg;
            h.deflate_off = 2;
            if (((cmf << 8) + flg) % 31 != 0) { h.status = 1; h.err = ERR_ZLIB_CHECK; h.a0 = cmf; h.a1 = flg; }
            else if ((cmf & 15) != 8) { h.status = 1; h.err = ERR_METHOD; h.a0 = cmf & 15; }
            else if ((cmf >> 4) > 7) { h.status = 1; h.err = ERR_CINFO; h.a0 = cmf >> 4; }
            else if (flg & 0x20) {
                if (n < 6) { h.status = 2; h.err = ERR_EOF; h.deflate_off = n; }
                else { h.status = 1; h.err = ERR_FDICT; h.deflate_off = 6;
                       h.a0 = (uint32_t)p[2] << 24 | (uint32_t)p[3] << 16 | (uint32_t)p[4] << 8 | p[5]; }
            }
        }
    } else if (format == 2) {
        uint64_t pos = 0;
        if (n < 10) { h.status = 2; h.err = ERR_EOF; pos = n; }
        else {
            pos = 10;
            const uint32_t flags = p[3];
            h.flags = flags;
            cf.flg = (uint8_t)flags; cf.xfl = p[8]; cf.os = p[9];
            cf.mtime = (uint32_t)p[4] | (uint32_t)p[5] << 8 | (uint32_t)p[6] << 16 | (uint32_t)p[7] << 24;
            if (p[0] != 31 || p[1] != 139) { h.status = 1; h.err = ERR_GZIP_ID; }
            else if (p[2] != 8) { h.status = 1; h.err = ERR_METHOD; h.a0 = p[2]; }
            else {
                if (flags & 4) {
                    if (n - pos < 2) { h.status = 2; h.err = ERR_EOF; pos = n; }
                    else {
                        const uint64_t xl = (uint64_t)p[pos] | (uint64_t)p[pos + 1] << 8;
                        pos += 2;
                        cf.extra_off = pos; cf.extra_len = xl;
                        uint64_t lim = xl, q = pos;
                        while (lim > 0 && h.status == 0) {
                            if (lim < 4 || n - q < 4) { h.status = 2; h.err = ERR_EOF; q = n; break; }
                            const uint64_t dl = (uint64_t)p[q + 2] | (uint64_t)p[q + 3] << 8;
                            q += 4; lim -= 4;
                            if (lim < dl || n - q < dl) { h.status = 2; h.err = ERR_EOF; q = n; break; }
                            q += dl; lim -= dl;
                        }
                        pos = q;
                    }
                }
                for (int k = 0; k < 2 && h.status == 0; ++k) {
                    if (!(flags & (k ? 16 : 8))) continue;
                    const uint64_t s = pos;
                    for (;;) {
                        if (pos >= n) { h.status = 2; h.err = ERR_EOF; break; }
                        if (p[pos++] == 0) break;
                    }
                    if (k) { cf.comment_off = s; cf.comment_len = pos - s; }
                    else { cf.name_off = s; cf.name_len = pos - s; }
                }
                if (h.status == 0 && (flags & 2)) {
                    if (n - pos < 2) { h.status = 2; h.err = ERR_EOF; pos = n; }
                    else {
                        const uint32_t crc = (uint32_t)p[pos] | (uint32_t)p[pos + 1] << 8;
                        pos += 2;
                        uint32_t c = 0xFFFFFFFFu;
                        auto upd = [&](uint32_t byte) {
                            c ^= byte;
                            for (int b = 0; b < 8; ++b) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
                        };
                        for (int k = 0; k < 10; ++k) {
                            uint32_t b = p[k];
                            if (k == 3) b = flags & (1 | 4 | 8 | 16);
                            if (k == 8) b = (b == 4 || b == 2) ? b : 0;
                            upd(b);
                        }
                        if (flags & 4) {
                            upd((uint32_t)cf.extra_len & 0xFF); upd((uint32_t)(cf.extra_len >> 8) & 0xFF);
                            for (uint64_t k = 0; k < cf.extra_len; ++k) upd(p[cf.extra_off + k]);
                        }
                        for (uint64_t j = 0; j < cf.name_len; ++j) upd(p[cf.name_off + j]);
                        for (uint64_t j = 0; j < cf.comment_len; ++j) upd(p[cf.comment_off + j]);
                        const uint32_t expect = (~c) & 0xFFFF;
                        if (crc != expect) { h.status = 1; h.err = ERR_HCRC; h.a0 = crc; h.a1 = expect; }
                    }
                }
            }
        }
        h.deflate_off = pos;
    }
    if (f) *f = cf;
    return h;
}

}  // namespace lfx
