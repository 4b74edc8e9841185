// cloud_io.cpp -- see cloud_io.hpp for the reference lines restated here.
#include "cloud_io.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>

namespace ccal_host {

double metric_scale(const std::string& metric)
{
    if (metric == "mm") return 1000.0;
    if (metric == "cm") return 100.0;
    return 1.0;
}

namespace {

std::string lower(std::string s)
{
    for (auto& c : s) c = (char)std::tolower((unsigned char)c);
    return s;
}

std::vector<std::string> split_ws(const std::string& line)
{
    std::vector<std::string> out;
    std::istringstream is(line);
    std::string t;
    while (is >> t) out.push_back(t);
    return out;
}

struct Prop {
    std::string name;
    int size = 0;       // bytes, 0 for list properties
    char kind = 'f';    // 'f' float, 'd' double, 'i' signed int, 'u' unsigned int
    bool is_list = false;
};

bool scalar_type(const std::string& t, int& size, char& kind)
{
    static const struct { const char* n; int s; char k; } tbl[] = {
        {"char", 1, 'i'}, {"int8", 1, 'i'}, {"uchar", 1, 'u'}, {"uint8", 1, 'u'},
        {"short", 2, 'i'}, {"int16", 2, 'i'}, {"ushort", 2, 'u'}, {"uint16", 2, 'u'},
        {"int", 4, 'i'}, {"int32", 4, 'i'}, {"uint", 4, 'u'}, {"uint32", 4, 'u'},
        {"float", 4, 'f'}, {"float32", 4, 'f'}, {"double", 8, 'd'}, {"float64", 8, 'd'}};
    for (auto& e : tbl)
        if (t == e.n) { size = e.s; kind = e.k; return true; }
    return false;
}

double load_scalar(const unsigned char* p, int size, char kind, bool swap)
{
    unsigned char b[8];
    std::memcpy(b, p, size);
    if (swap) std::reverse(b, b + size);
    switch (kind) {
    case 'f': { float v; std::memcpy(&v, b, 4); return v; }
    case 'd': { double v; std::memcpy(&v, b, 8); return v; }
    case 'i':
        if (size == 1) { int8_t v; std::memcpy(&v, b, 1); return v; }
        if (size == 2) { int16_t v; std::memcpy(&v, b, 2); return v; }
        { int32_t v; std::memcpy(&v, b, 4); return v; }
    default:
        if (size == 1) { uint8_t v; std::memcpy(&v, b, 1); return v; }
        if (size == 2) { uint16_t v; std::memcpy(&v, b, 2); return v; }
        { uint32_t v; std::memcpy(&v, b, 4); return v; }
    }
}

bool host_is_little_endian()
{
    const uint16_t one = 1;
    return *reinterpret_cast<const unsigned char*>(&one) == 1;
}

// A byte buffer that is not cleared before the file is read into it.
class RawBytes {
public:
    explicit RawBytes(size_t n) : p_(new unsigned char[n ? n : 1]), n_(n) {}
    unsigned char* data() const { return p_.get(); }
    size_t size() const { return n_; }
private:
    std::unique_ptr<unsigned char[]> p_;
    size_t n_;
};

// x, y, z of `count` fixed-size records (or of three separate columns: any strides) into out.  Plain
// 4-byte floats in host byte order -- what every depth camera writes -- are copied; anything else goes
// through the generic scalar loader.
struct Column { const unsigned char* base; size_t stride; int size; char kind; };
void gather_xyz(const Column c[3], size_t count, bool swap, Cloud& out)
{
    const size_t at = out.size();
    out.resize(at + 3 * count);
    float* dst = out.data() + at;
    const bool plain = !swap && c[0].size == 4 && c[1].size == 4 && c[2].size == 4 && c[0].kind == 'f' &&
                       c[1].kind == 'f' && c[2].kind == 'f';
    if (plain) {
        const unsigned char *px = c[0].base, *py = c[1].base, *pz = c[2].base;
        for (size_t r = 0; r < count; ++r, px += c[0].stride, py += c[1].stride, pz += c[2].stride) {
            std::memcpy(dst + 3 * r + 0, px, 4);
            std::memcpy(dst + 3 * r + 1, py, 4);
            std::memcpy(dst + 3 * r + 2, pz, 4);
        }
        return;
    }
    for (size_t r = 0; r < count; ++r)
        for (int k = 0; k < 3; ++k)
            dst[3 * r + k] = (float)load_scalar(c[k].base + r * c[k].stride, c[k].size, c[k].kind, swap);
}

// strtof's value for the number at p, without its cost in the common case.  A decimal with at most 19
// significant digits and a decimal exponent within +-22 is one exact integer times or over one exact power
// of ten: a single correctly rounded double operation (Clinger's fast path).  Rounding that double to
// float is the correctly rounded float of the decimal unless the double sits within one unit of a
// midpoint between two floats, where the two roundings could disagree; that case, and everything
// unusual (more digits, large exponents, inf/nan/hex, no digits at all), is left to strtof itself.
float parse_float(const char* p, char** end)
{
    static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                      1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* q = p;
    while (*q == ' ' || (*q >= '\t' && *q <= '\r')) ++q;
    const char* start = q;
    bool neg = false;
    if (*q == '+' || *q == '-') neg = *q++ == '-';
    uint64_t m = 0;
    int digits = 0, dropped_exp = 0, frac = 0;
    bool any = false;
    for (; *q >= '0' && *q <= '9'; ++q) {
        any = true;
        if (digits < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++digits; }
        else ++dropped_exp, digits = 20;
    }
    if (*q == '.') {
        ++q;
        for (; *q >= '0' && *q <= '9'; ++q) {
            any = true;
            if (digits < 19) { m = m * 10 + (uint64_t)(*q - '0'); if (m) ++digits; ++frac; }
            else digits = 20;
        }
    }
    if (!any || digits > 19) return std::strtof(p, end);
    int e10 = dropped_exp - frac;
    if (*q == 'e' || *q == 'E') {
        const char* r = q + 1;
        bool eneg = false;
        if (*r == '+' || *r == '-') eneg = *r++ == '-';
        if (*r >= '0' && *r <= '9') {
            int ev = 0;
            for (; *r >= '0' && *r <= '9'; ++r) ev = ev < 10000 ? ev * 10 + (*r - '0') : ev;
            e10 += eneg ? -ev : ev;
            q = r;
        }
    } else if (*q == 'x' || *q == 'X' || *q == 'p' || *q == 'P') {
        return std::strtof(p, end);          // a hexadecimal float ("0x...")
    }
    if (m > ((uint64_t)1 << 53) || e10 < -22 || e10 > 22) return std::strtof(p, end);
    double d = (double)m;
    d = e10 < 0 ? d / kPow10[-e10] : d * kPow10[e10];
    uint64_t bits;
    std::memcpy(&bits, &d, 8);
    const uint64_t low = bits & (((uint64_t)1 << 29) - 1);          // the bits a float drops
    const uint64_t half = (uint64_t)1 << 28;
    const int biased = (int)((bits >> 52) & 0x7ff);
    // float's normal range only (below it the dropped bits are more), and not next to a midpoint
    if (m != 0 && (biased < 1023 - 126 || biased > 1023 + 127 || (low >= half - 1 && low <= half + 1)))
        return std::strtof(p, end);
    (void)start;
    if (end) *end = const_cast<char*>(q);
    const float f = (float)d;
    return neg ? -f : f;
}

// The rest of a stream as one string (a terminating NUL follows it, as c_str() guarantees).
std::string slurp(std::ifstream& f)
{
    std::string body;
    const std::streampos at = f.tellg();
    f.seekg(0, std::ios::end);
    const std::streampos end = f.tellg();
    f.seekg(at);
    if (end > at) {
        body.resize((size_t)(end - at));
        f.read(&body[0], (std::streamsize)body.size());
        body.resize((size_t)f.gcount());
    }
    return body;
}

// One text line of whitespace-separated columns: the floats of the three wanted columns.  Lines end at
// '\n' (a '\r' before it counts as whitespace).  Returns the start of the next line, or nullptr when the
// line has fewer columns than `need`.
const char* parse_columns(const char* p, const char* body_end, const int want[3], int need, float v[3])
{
    int col = 0;
    while (true) {
        while (p < body_end && *p != '\n' && (*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
        if (p >= body_end || *p == '\n') break;
        const char* tok = p;
        while (p < body_end && !(*p == ' ' || (*p >= '\t' && *p <= '\r'))) ++p;
        for (int k = 0; k < 3; ++k)
            if (want[k] == col) v[k] = parse_float(tok, nullptr);
        ++col;
    }
    if (col < need) return nullptr;
    return p < body_end ? p + 1 : p;
}

} // namespace

// ---------------------------------------------------------------------------
// PLY (ascii, binary_little_endian, binary_big_endian); vertex x,y,z of any
// scalar type, other elements skipped.
// ---------------------------------------------------------------------------
Cloud read_ply(const std::string& filename)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + filename);
    std::string line;
    if (!std::getline(f, line) || split_ws(line).empty() || split_ws(line)[0] != "ply")
        throw std::runtime_error(filename + ": not a PLY file");
    struct Element { std::string name; size_t count = 0; std::vector<Prop> props; };
    std::vector<Element> elems;
    std::string format;
    while (std::getline(f, line)) {
        auto t = split_ws(line);
        if (t.empty()) continue;
        if (t[0] == "end_header") break;
        if (t[0] == "format" && t.size() >= 2) format = t[1];
        else if (t[0] == "element" && t.size() >= 3) {
            Element e;
            e.name = t[1];
            e.count = (size_t)std::strtoull(t[2].c_str(), nullptr, 10);
            elems.push_back(e);
        } else if (t[0] == "property" && !elems.empty()) {
            Prop p;
            if (t.size() >= 5 && t[1] == "list") { p.is_list = true; p.name = t[4]; }
            else if (t.size() >= 3) {
                if (!scalar_type(t[1], p.size, p.kind))
                    throw std::runtime_error(filename + ": unknown PLY type " + t[1]);
                p.name = t[2];
            }
            elems.back().props.push_back(p);
        }
    }
    if (format != "ascii" && format != "binary_little_endian" && format != "binary_big_endian")
        throw std::runtime_error(filename + ": unsupported PLY format '" + format + "'");
    const bool ascii = format == "ascii";
    const bool swap = !ascii && ((format == "binary_little_endian") != host_is_little_endian());

    Cloud out;
    for (const Element& e : elems) {
        int ix = -1, iy = -1, iz = -1, rec = 0;
        bool has_list = false;
        std::vector<int> offs;
        for (size_t i = 0; i < e.props.size(); ++i) {
            offs.push_back(rec);
            if (e.props[i].is_list) has_list = true;
            rec += e.props[i].size;
            if (e.props[i].name == "x") ix = (int)i;
            if (e.props[i].name == "y") iy = (int)i;
            if (e.props[i].name == "z") iz = (int)i;
        }
        const bool is_vertex = e.name == "vertex";
        if (is_vertex && (ix < 0 || iy < 0 || iz < 0))
            throw std::runtime_error(filename + ": vertex element without x/y/z");
        if (ascii) {
            if (!is_vertex) {
                for (size_t r = 0; r < e.count; ++r)
                    if (!std::getline(f, line)) throw std::runtime_error(filename + ": truncated PLY body");
                continue;
            }
            const std::string body = slurp(f);
            const char *p = body.c_str(), *body_end = p + body.size();
            const int want[3] = {ix, iy, iz};
            const int need = std::max(ix, std::max(iy, iz)) + 1;
            out.resize(3 * e.count);
            for (size_t r = 0; r < e.count; ++r) {
                if (p >= body_end) throw std::runtime_error(filename + ": truncated PLY body");
                p = parse_columns(p, body_end, want, need, out.data() + 3 * r);
                if (!p) throw std::runtime_error(filename + ": short vertex line");
            }
        } else {
            if (has_list) {
                if (is_vertex) throw std::runtime_error(filename + ": list property in a binary vertex element");
                // a variable-size element after the vertices cannot be skipped blindly,
                // but nothing after it is needed either
                if (!out.empty()) break;
                throw std::runtime_error(filename + ": list element before the vertices");
            }
            const RawBytes buf((size_t)rec * e.count);
            f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
            if ((size_t)f.gcount() != buf.size()) throw std::runtime_error(filename + ": truncated PLY body");
            if (!is_vertex) continue;
            const Column cols[3] = {{buf.data() + offs[ix], (size_t)rec, e.props[ix].size, e.props[ix].kind},
                                    {buf.data() + offs[iy], (size_t)rec, e.props[iy].size, e.props[iy].kind},
                                    {buf.data() + offs[iz], (size_t)rec, e.props[iz].size, e.props[iz].kind}};
            gather_xyz(cols, e.count, swap, out);
        }
        if (is_vertex) break;
    }
    return out;
}

// LZF decompression (the codec of PCD "binary_compressed" bodies; format of liblzf 3.x): a control
// byte below 32 starts a literal run of ctrl + 1 bytes; otherwise the top three bits give a length
// (7 = one more length byte follows), the low five bits and the next byte a distance, and
// length + 2 bytes are copied from that far back in the output.  Returns false on a malformed stream.
static bool lzf_decompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len)
{
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        const unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            std::memcpy(out + op, in + ip, run);
            ip += run;
            op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) {
                if (ip >= in_len) return false;
                len += in[ip++];
            }
            if (ip >= in_len) return false;
            const size_t dist = ((size_t)(ctrl & 0x1f) << 8) + in[ip++] + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t k = 0; k < len; ++k, ++op) out[op] = out[op - dist];   // ranges may overlap
        }
    }
    return op == out_len;
}

// ---------------------------------------------------------------------------
// PCD v0.7 (DATA ascii | binary | binary_compressed), generic field table
// ---------------------------------------------------------------------------
Cloud read_pcd(const std::string& filename)
{
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + filename);
    std::vector<std::string> fields, types;
    std::vector<int> sizes, counts;
    size_t points = 0, width = 0, height = 1;
    bool have_points = false;
    std::string data, line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        auto t = split_ws(line);
        if (t.empty() || t[0][0] == '#') continue;
        const std::string key = t[0];
        if (key == "FIELDS") fields.assign(t.begin() + 1, t.end());
        else if (key == "SIZE") for (size_t i = 1; i < t.size(); ++i) sizes.push_back(std::atoi(t[i].c_str()));
        else if (key == "TYPE") types.assign(t.begin() + 1, t.end());
        else if (key == "COUNT") for (size_t i = 1; i < t.size(); ++i) counts.push_back(std::atoi(t[i].c_str()));
        else if (key == "WIDTH" && t.size() > 1) width = (size_t)std::strtoull(t[1].c_str(), nullptr, 10);
        else if (key == "HEIGHT" && t.size() > 1) height = (size_t)std::strtoull(t[1].c_str(), nullptr, 10);
        else if (key == "POINTS" && t.size() > 1) { points = (size_t)std::strtoull(t[1].c_str(), nullptr, 10); have_points = true; }
        else if (key == "DATA" && t.size() > 1) { data = lower(t[1]); break; }
    }
    if (!have_points) points = width * height;
    else if (width && height && points != width * height)
        throw std::runtime_error(filename + ": PCD POINTS differs from WIDTH x HEIGHT");
    if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size())
        throw std::runtime_error(filename + ": malformed PCD header");
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (counts.size() != fields.size()) throw std::runtime_error(filename + ": malformed PCD COUNT");
    int ix = -1, iy = -1, iz = -1, rec = 0, col = 0;
    std::vector<int> offs, cols;
    for (size_t i = 0; i < fields.size(); ++i) {
        offs.push_back(rec);
        cols.push_back(col);
        rec += sizes[i] * counts[i];
        col += counts[i];
        if (fields[i] == "x") ix = (int)i;
        if (fields[i] == "y") iy = (int)i;
        if (fields[i] == "z") iz = (int)i;
    }
    if (ix < 0 || iy < 0 || iz < 0) throw std::runtime_error(filename + ": PCD without x/y/z fields");
    auto kind_of = [&](int i) -> char {
        const char t = types[i].empty() ? 'F' : types[i][0];
        if (t == 'F') return sizes[i] == 8 ? 'd' : 'f';
        return t == 'I' ? 'i' : 'u';
    };
    Cloud out;
    if (data == "ascii") {
        const std::string body = slurp(f);
        const char *p = body.c_str(), *body_end = p + body.size();
        const int want[3] = {cols[ix], cols[iy], cols[iz]};
        out.resize(points * 3);
        for (size_t r = 0; r < points; ++r) {
            if (p >= body_end) throw std::runtime_error(filename + ": truncated PCD body");
            p = parse_columns(p, body_end, want, col, out.data() + 3 * r);
            if (!p) throw std::runtime_error(filename + ": short PCD line");
        }
    } else if (data == "binary") {
        const RawBytes buf((size_t)rec * points);
        f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size());
        if ((size_t)f.gcount() != buf.size()) throw std::runtime_error(filename + ": truncated PCD body");
        const bool swap = !host_is_little_endian();
        const Column c3[3] = {{buf.data() + offs[ix], (size_t)rec, sizes[ix], kind_of(ix)},
                              {buf.data() + offs[iy], (size_t)rec, sizes[iy], kind_of(iy)},
                              {buf.data() + offs[iz], (size_t)rec, sizes[iz], kind_of(iz)}};
        gather_xyz(c3, points, swap, out);
    } else if (data == "binary_compressed") {
        // PCL's writer: uint32 compressed size, uint32 uncompressed size, an LZF stream; the
        // uncompressed body holds the fields one after the other (all points' values of field 0,
        // then of field 1, ...), not point records
        unsigned char sz[8];
        f.read(reinterpret_cast<char*>(sz), 8);
        if (f.gcount() != 8) throw std::runtime_error(filename + ": truncated PCD body");
        auto u32 = [](const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
        const size_t n_comp = u32(sz), n_raw = u32(sz + 4);
        if (n_raw != (size_t)rec * points) throw std::runtime_error(filename + ": PCD compressed body has the wrong size");
        const RawBytes comp(n_comp), buf(n_raw);
        f.read(reinterpret_cast<char*>(comp.data()), (std::streamsize)n_comp);
        if ((size_t)f.gcount() != n_comp) throw std::runtime_error(filename + ": truncated PCD body");
        if (!lzf_decompress(comp.data(), n_comp, buf.data(), n_raw))
            throw std::runtime_error(filename + ": malformed LZF stream in the PCD body");
        const bool swap = !host_is_little_endian();
        // field i starts at points * (bytes of the fields before it)
        auto column = [&](int i) {
            return Column{buf.data() + points * (size_t)offs[i], (size_t)(sizes[i] * counts[i]), sizes[i], kind_of(i)};
        };
        const Column c3[3] = {column(ix), column(iy), column(iz)};
        gather_xyz(c3, points, swap, out);
    } else {
        throw std::runtime_error(filename + ": unsupported PCD DATA '" + data + "'");
    }
    return out;
}

// src/helpers.cpp:168-194: skip the first line; each following non-empty line is
// "<token> x,y,z ..."; coordinates = stof / scale.
Cloud read_xyz_text(const std::string& filename, double scale)
{
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("cannot open " + filename);
    std::string line;
    std::getline(f, line);
    Cloud out;
    while (std::getline(f, line) && !line.empty()) {
        size_t sp = line.find(' ');
        if (sp == std::string::npos) throw std::runtime_error(filename + ": malformed xyz line");
        size_t sp2 = line.find(' ', sp + 1);
        std::string tok = line.substr(sp + 1, sp2 == std::string::npos ? std::string::npos : sp2 - sp - 1);
        float v[3];
        size_t pos = 0;
        for (int k = 0; k < 3; ++k) {
            size_t comma = tok.find(',', pos);
            std::string piece = tok.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
            char* endp = nullptr;
            float s = parse_float(piece.c_str(), &endp);
            if (endp == piece.c_str()) throw std::runtime_error(filename + ": malformed xyz coordinate");
            v[k] = (float)((double)s / scale);
            pos = comma == std::string::npos ? tok.size() : comma + 1;
        }
        out.insert(out.end(), v, v + 3);
    }
    return out;
}

// src/helpers.cpp:108-207
Cloud read_point_cloud(const std::string& filename, const std::string& metric)
{
    const double scale = metric_scale(metric);
    size_t dot = filename.rfind('.');
    const std::string type = lower(dot == std::string::npos ? filename : filename.substr(dot + 1));
    Cloud c;
    if (type == "pcd") c = read_pcd(filename);
    else if (type == "ply") c = read_ply(filename);
    else if (type == "xyz") return read_xyz_text(filename, scale);
    else throw std::runtime_error("Format not supported: " + filename);
    if (metric != "m")
        for (auto& v : c) v = (float)((double)v / scale);
    return c;
}

Cloud filter_z(const Cloud& in, double z_max)
{
    Cloud out;
    out.reserve(in.size());
    for (size_t i = 0; i + 2 < in.size(); i += 3)
        if ((double)in[i + 2] <= z_max) out.insert(out.end(), in.begin() + i, in.begin() + i + 3);
    return out;
}

Cloud voxel_grid(const Cloud& in, float leaf)
{
    const size_t n = in.size() / 3;
    Cloud out;
    if (n == 0) return out;
    const float inv = 1.0f / leaf;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k) {
            const float v = in[3 * i + k];
            if (!std::isfinite(v)) continue;
            mn[k] = std::min(mn[k], v);
            mx[k] = std::max(mx[k], v);
        }
    int64_t minb[3], div[3];
    for (int k = 0; k < 3; ++k) {
        minb[k] = (int64_t)std::floor(mn[k] * inv);
        div[k] = (int64_t)std::floor(mx[k] * inv) - minb[k] + 1;
    }
    struct Rec { int64_t key; uint32_t idx; };
    std::vector<Rec> rec;
    rec.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        const float* p = &in[3 * i];
        if (!std::isfinite(p[0]) || !std::isfinite(p[1]) || !std::isfinite(p[2])) continue;
        const int64_t i0 = (int64_t)(std::floor(p[0] * inv) - (float)minb[0]);
        const int64_t i1 = (int64_t)(std::floor(p[1] * inv) - (float)minb[1]);
        const int64_t i2 = (int64_t)(std::floor(p[2] * inv) - (float)minb[2]);
        rec.push_back({i0 + i1 * div[0] + i2 * div[0] * div[1], (uint32_t)i});
    }
    std::stable_sort(rec.begin(), rec.end(), [](const Rec& a, const Rec& b) { return a.key < b.key; });
    for (size_t i = 0; i < rec.size();) {
        size_t j = i;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (; j < rec.size() && rec[j].key == rec[i].key; ++j) {
            const float* p = &in[3 * (size_t)rec[j].idx];
            sx += p[0]; sy += p[1]; sz += p[2];
        }
        const float cnt = (float)(j - i);
        out.push_back(sx / cnt);
        out.push_back(sy / cnt);
        out.push_back(sz / cnt);
        i = j;
    }
    return out;
}

// src/camera_calibration_optimization.cpp:81-93 and src/helpers.cpp:244-251
int parse_pose_csv(const std::string& line, double scale, bool angle_is_degree, double pose[6])
{
    int n = 0;
    size_t pos = 0;
    while (pos <= line.size() && n < 6) {
        size_t comma = line.find(',', pos);
        std::string piece = line.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        char* endp = nullptr;
        const float f = std::strtof(piece.c_str(), &endp);  // std::stof
        if (endp == piece.c_str()) break;
        if (n < 3) pose[n] = f / scale;
        else {
            double value = f;
            if (angle_is_degree) value = (int)value * 3.141592 / 180;  // degreeToRadian(int)
            pose[n] = value;
        }
        ++n;
        if (comma == std::string::npos) break;
        pos = comma + 1;
    }
    return n;
}

std::vector<std::vector<double>> read_transformations(const std::string& filename,
                                                      const std::string& metric)
{
    std::ifstream f(filename);
    if (!f) throw std::runtime_error("cannot open " + filename);
    const double scale = metric_scale(metric);
    std::vector<std::vector<double>> out;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) break;   // `while(getline(file,trans)&&trans.size())`
        double pose[6] = {0, 0, 0, 0, 0, 0};
        if (parse_pose_csv(line, scale, false, pose) < 6)
            throw std::runtime_error(filename + ": a transformation needs 6 comma-separated values");
        out.emplace_back(pose, pose + 6);
    }
    return out;
}

// src/camera_calibration_optimization.cpp:140-146
int file_id(const std::string& filename)
{
    size_t slash = filename.rfind('/');
    std::string file = slash == std::string::npos ? filename : filename.substr(slash + 1);
    file = file.substr(0, file.find('.'));
    size_t us = file.find('_');
    if (us == std::string::npos) throw std::runtime_error("no '_<id>' in cloud file name " + filename);
    std::string id = file.substr(us + 1);
    id = id.substr(0, id.find('_'));
    try {
        return std::stoi(id);
    } catch (...) {
        throw std::runtime_error("no numeric id in cloud file name " + filename);
    }
}

} // namespace ccal_host
