// fits_frame.cpp -- the FITS framing around a payload, host side of the C ABI (include/nlstack.h): header cards,
// END, 2880-byte blocks.  What crosses PCIe and is decoded / encoded on the device is the payload
// (nl_stack_upload_frame_fits, nl_stack_download_result_fits, csrc/ingest.hip); this file finds it in a file image and
// frames a result for writing, so that a host without the Go side (tests/test_c1_plumbing.py) goes file -> device ->
// file through the product alone.
//   header reader   internal/fits/read.go:445-469 (2880-byte units, 80-byte cards, the line grammar of :525-559),
//                   :97-147 (SIMPLE, BITPIX, NAXIS, NAXISn mandatory; BZERO, BSCALE, EXPOSURE / EXPTIME optional)
//   header writer   internal/fits/write.go:54-89 (the cards Image.Write emits, END, padding with spaces),
//                   :104-147 ("%-8s= %20s / %-47s", %g of a float32)
#include <errno.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "../../include/nlstack.h"

namespace nl { void set_last_error(const char *msg); }

namespace {

constexpr int64_t kBlock = 2880;
constexpr int kCard = 80;

int fail(int code, const std::string &msg)
{
    nl::set_last_error(msg.c_str());
    return code;
}

bool is_key_char(char c) { return (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '_' || c == '-'; }
bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

enum Kind { kNone, kBool, kInt, kFloat, kOther };
struct Card { std::string key; Kind kind = kNone; bool b = false; int32_t i = 0; float f = 0.0f; bool end = false; };

// one 80-byte card through the grammar of read.go:525-559 (the regular expression's alternatives in its order: blank,
// HISTORY, COMMENT, key = value [/ comment], END); what does not parse is ignored there with a warning, here silently
Card parse_card(const char *line)
{
    Card c;
    int n = kCard;
    // END followed by white space only
    if (n >= 3 && memcmp(line, "END", 3) == 0) {
        bool blank = true;
        for (int k = 3; k < n; k++) blank = blank && is_space(line[k]);
        if (blank) { c.end = true; return c; }
    }
    int p = 0;
    while (p < n && is_key_char(line[p])) p++;
    if (p == 0) return c;
    const std::string key(line, (size_t)p);
    while (p < n && is_space(line[p])) p++;
    if (p >= n || line[p] != '=') return c;
    p++;
    while (p < n && is_space(line[p])) p++;
    // value: [TF] | [+-]?[0-9]+ | [+-]?[0-9]*\.[0-9]*([ED][-+]?[0-9]+)? | '...' ; then white space, optional / comment, end
    auto tail_ok = [&](int q) {
        while (q < n && is_space(line[q])) q++;
        return q == n || line[q] == '/';
    };
    if (p < n && (line[p] == 'T' || line[p] == 'F') && tail_ok(p + 1)) {
        c.key = key; c.kind = kBool; c.b = line[p] == 'T';
        return c;
    }
    int q = p;
    if (q < n && (line[q] == '+' || line[q] == '-')) q++;
    const int d0 = q;
    while (q < n && line[q] >= '0' && line[q] <= '9') q++;
    if (q > d0 && tail_ok(q)) {                              // integer (strconv.ParseInt, then int32(): wraps)
        const std::string txt(line + p, (size_t)(q - p));
        errno = 0;
        const long long v = strtoll(txt.c_str(), nullptr, 10);
        if (errno == 0) { c.key = key; c.kind = kInt; c.i = (int32_t)v; }
        return c;
    }
    if (q < n && line[q] == '.') {                           // float: digits '.' digits, optional exponent
        q++;
        while (q < n && line[q] >= '0' && line[q] <= '9') q++;
        bool d_exp = false;
        if (q < n && (line[q] == 'E' || line[q] == 'D')) {
            int r = q + 1;
            if (r < n && (line[r] == '+' || line[r] == '-')) r++;
            const int e0 = r;
            while (r < n && line[r] >= '0' && line[r] <= '9') r++;
            if (r > e0) { d_exp = line[q] == 'D'; q = r; }
        }
        if (tail_ok(q)) {
            // strconv.ParseFloat does not know the D exponent: such a value matches the grammar and is then dropped
            if (!d_exp && !(q - p == 1) && !(q - p == 2 && (line[p] == '+' || line[p] == '-'))) {
                const std::string txt(line + p, (size_t)(q - p));
                c.key = key; c.kind = kFloat; c.f = (float)strtod(txt.c_str(), nullptr);
            }
            return c;
        }
    }
    if (p < n && line[p] == '\'') {                          // string (not needed for the geometry): recognised, not kept
        int r = p + 1;
        while (r < n && line[r] != '\'') r++;
        if (r < n && tail_ok(r + 1)) { c.key = key; c.kind = kOther; }
    }
    return c;
}

// fmt's %g of a float32: shortest digits that round-trip; %e form (d.ddde+XX) when the decimal exponent is < -4 or >= 6
std::string go_g(float v)
{
    if (v != v) return "NaN";
    if (isinf(v)) return v > 0 ? "+Inf" : "-Inf";
    if (v == 0.0f) return signbit(v) ? "-0" : "0";
    char buf[64];
    int prec = 1;
    for (; prec <= 9; prec++) {
        snprintf(buf, sizeof buf, "%.*e", prec - 1, (double)v);
        if (strtof(buf, nullptr) == v) break;
    }
    // buf = [-]d[.ddd]e[+-]XX
    std::string s(buf);
    const size_t epos = s.find('e');
    const int exp = atoi(s.c_str() + epos + 1);
    std::string mant = s.substr(0, epos);
    std::string sign;
    if (mant[0] == '-') { sign = "-"; mant = mant.substr(1); }
    std::string digits;
    for (char ch : mant) if (ch != '.') digits += ch;
    while (digits.size() > 1 && digits.back() == '0') digits.pop_back();
    // strconv's %e decision for the SHORTEST form uses a precision of 6 whatever the number of digits (ftoa.go: `if
    // shortest { eprec = 6 }` comes after the digit-count adjustment): float32(1234567) -> 1.234567e+06
    if (exp < -4 || exp >= 6) {
        std::string out = sign + digits.substr(0, 1);
        if (digits.size() > 1) out += "." + digits.substr(1);
        char e[16];
        snprintf(e, sizeof e, "e%c%02d", exp < 0 ? '-' : '+', abs(exp));
        return out + e;
    }
    if (exp >= 0) {
        std::string out = sign;
        if ((int)digits.size() <= exp + 1) out += digits + std::string((size_t)(exp + 1 - (int)digits.size()), '0');
        else out += digits.substr(0, (size_t)exp + 1) + "." + digits.substr((size_t)exp + 1);
        return out;
    }
    return sign + "0." + std::string((size_t)(-exp - 1), '0') + digits;
}

void put_card(std::string *sb, const char *key, const std::string &value, const char *comment)
{
    char line[96];
    snprintf(line, sizeof line, "%-8.8s= %20s / %-47.47s", key, value.c_str(), comment);
    sb->append(line);
}

}  // namespace

extern "C" {

int64_t nl_fits_padded_bytes(int64_t payload_bytes)
{
    return payload_bytes <= 0 ? 0 : (payload_bytes + kBlock - 1) / kBlock * kBlock;
}

int nl_fits_parse_header(const void *file_bytes, int64_t n_bytes, int id, nl_fits_header_t *out)
{
    if (!file_bytes || !out) return fail(NL_ERR_INVALID_ARG, "fits_parse_header: null argument");
    memset(out, 0, sizeof *out);
    const char *base = static_cast<const char *>(file_bytes);
    bool simple = false, have_bitpix = false, have_naxis = false, end = false;
    bool have_n[NL_FITS_MAX_AXES] = {};
    bool have_exposure = false, have_exptime = false;
    float exptime = 0.0f;
    out->bscale = 1.0f;
    int64_t length = 0;
    while (!end) {                                           // read.go:448-467: whole 2880-byte units until END
        if (length + kBlock > n_bytes)
            return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": unexpected EOF");
        for (int l = 0; l < (int)(kBlock / kCard) && !end; l++) {
            const Card c = parse_card(base + length + (int64_t)l * kCard);
            if (c.end) { end = true; break; }
            if (c.kind == kNone) continue;
            const float num = c.kind == kInt ? (float)c.i : c.f;
            const bool numeric = c.kind == kInt || c.kind == kFloat;
            if (c.key == "SIMPLE" && c.kind == kBool) simple = c.b;
            else if (c.key == "BITPIX" && c.kind == kInt) { out->bitpix = c.i; have_bitpix = true; }
            else if (c.key == "NAXIS" && c.kind == kInt) { out->naxis = c.i; have_naxis = true; }
            else if (c.key.size() > 5 && c.key.compare(0, 5, "NAXIS") == 0 && c.kind == kInt) {
                const int ax = atoi(c.key.c_str() + 5);
                if (ax >= 1 && ax <= NL_FITS_MAX_AXES && c.key == "NAXIS" + std::to_string(ax)) { out->naxisn[ax - 1] = c.i; have_n[ax - 1] = true; }
            }
            else if (c.key == "BZERO" && numeric) out->bzero = num;
            else if (c.key == "BSCALE" && numeric) out->bscale = num;
            else if (c.key == "EXPOSURE" && numeric) { out->exposure = num; have_exposure = true; }
            else if (c.key == "EXPTIME" && numeric) { exptime = num; have_exptime = true; }
        }
        length += kBlock;
    }
    if (!simple) return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": Not a valid FITS file; SIMPLE=T missing in header");     // read.go:103-105
    if (!have_bitpix) return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": FITS header does not contain key BITPIX");
    if (!have_naxis) return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": FITS header does not contain key NAXIS");
    if (out->naxis < 0 || out->naxis > NL_FITS_MAX_AXES)
        return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": NAXIS " + std::to_string(out->naxis) + " not in [0, " + std::to_string(NL_FITS_MAX_AXES) + "]");
    // (the reference trusts these fields and sizes its buffers from them; a C ABI fed with untrusted files does not:
    // the BITPIX values read.go:172-445 decodes, no negative axis, no overflow of the byte counts)
    switch (out->bitpix) {
    case 8: case 16: case 32: case 64: case -32: case -64: break;
    default: return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": unsupported BITPIX " + std::to_string(out->bitpix));
    }
    int64_t pixels = 1;
    for (int a = 0; a < out->naxis; a++) {
        if (!have_n[a]) return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": FITS header does not contain key NAXIS" + std::to_string(a + 1));
        if (out->naxisn[a] < 0)
            return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": negative NAXIS" + std::to_string(a + 1) + " " + std::to_string(out->naxisn[a]));
        if (__builtin_mul_overflow(pixels, (int64_t)out->naxisn[a], &pixels))
            return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": the NAXISn of the header overflow 63 bits");
    }
    if (!have_exposure && have_exptime) out->exposure = exptime;                 // read.go:135-139
    const int64_t bytes_per = out->bitpix < 0 ? -out->bitpix / 8 : out->bitpix / 8;
    int64_t payload = 0;
    if (__builtin_mul_overflow(pixels, bytes_per, &payload) || payload > INT64_MAX - kBlock)
        return fail(NL_ERR_INVALID_ARG, std::to_string(id) + ": the payload size of the header overflows 63 bits");
    out->pixels = pixels;
    out->header_bytes = length;
    out->payload_bytes = payload;
    out->padded_payload_bytes = nl_fits_padded_bytes(out->payload_bytes);
    return NL_OK;
}

int64_t nl_fits_write_header(void *dst, int64_t capacity, int naxis, const int32_t *naxisn, float bzero, float bscale,
                             float exposure)
{
    if (naxis < 0 || naxis > NL_FITS_MAX_AXES || (naxis > 0 && !naxisn)) {
        fail(NL_ERR_INVALID_ARG, "fits_write_header: bad axes");
        return -1;
    }
    std::string sb;                                                               // write.go:56-71
    put_card(&sb, "SIMPLE", "T", "    FITS standard 4.0");
    put_card(&sb, "BITPIX", "-32", "    32-bit floating point");
    put_card(&sb, "NAXIS", std::to_string(naxis), "[1] Number of array dimensions");
    for (int a = 0; a < naxis; a++) {
        const std::string key = "NAXIS" + std::to_string(a + 1);
        put_card(&sb, key.c_str(), std::to_string(naxisn[a]), "[1] Array dimension");
    }
    put_card(&sb, "BZERO", go_g(bzero), "[1] Zero offset");
    put_card(&sb, "BSCALE", go_g(bscale), "[1] Data scale");
    if (exposure != 0) put_card(&sb, "EXPOSURE", go_g(exposure), "[s] Exposure duration");
    {
        char line[96];                                                            // writeString, write.go:150-171 (value of <= 18 characters)
        const char *value = "nightlight";
        snprintf(line, sizeof line, "%-8s= '%s'%*s / %-47.47s", "PROGRAM", value, (int)(18 - strlen(value)), "",
                 "    https://github.com/mlnoga/nightlight");
        sb.append(line);
    }
    sb.append("END");
    sb.append(std::string(kCard - 3, ' '));
    if (sb.size() % kBlock) sb.append(std::string((size_t)(kBlock - (int64_t)sb.size() % kBlock), ' '));      // :76-82
    if (!dst) return (int64_t)sb.size();
    if ((int64_t)sb.size() > capacity) {
        fail(NL_ERR_INVALID_ARG, "fits_write_header: buffer too small");
        return -1;
    }
    memcpy(dst, sb.data(), sb.size());
    return (int64_t)sb.size();
}

}  // extern "C"
