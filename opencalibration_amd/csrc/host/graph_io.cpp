// graph.json / surface PLY / checkpoint directory (see graph_io.hpp).  RapidJSON, OpenCV's PNG codec and the
// reference's base64.h are not in this image: the JSON reader is a cursor over the text that dispatches on member names
// (the reference looks members up by name, so member order does not matter there either), the writer reproduces
// PrettyWriter's layout, base64 is the standard alphabet with '=' padding.
#include "graph_io.hpp"

#include <algorithm>
#include <charconv>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <set>
#include <sstream>

namespace opencalibration_amd
{
namespace
{

// ---- base64 ------------------------------------------------------------------------------------------------------
const char B64[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";

std::string base64_encode(const unsigned char *data, size_t n)
{
    std::string out;
    out.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 2 < n; i += 3)
    {
        out += B64[data[i] >> 2];
        out += B64[((data[i] & 3) << 4) | (data[i + 1] >> 4)];
        out += B64[((data[i + 1] & 15) << 2) | (data[i + 2] >> 6)];
        out += B64[data[i + 2] & 63];
    }
    if (i < n)
    {
        out += B64[data[i] >> 2];
        if (i + 1 == n)
        {
            out += B64[(data[i] & 3) << 4];
            out += '=';
        }
        else
        {
            out += B64[((data[i] & 3) << 4) | (data[i + 1] >> 4)];
            out += B64[(data[i + 1] & 15) << 2];
        }
        out += '=';
    }
    return out;
}

// decodes up to the first character outside the alphabet (padding included), as Base64decode does
std::string base64_decode(const std::string &in)
{
    static int8_t lut[256];
    static bool init = false;
    if (!init)
    {
        std::fill(lut, lut + 256, (int8_t)-1);
        for (int i = 0; i < 64; i++)
            lut[(unsigned char)B64[i]] = (int8_t)i;
        init = true;
    }
    std::string out;
    out.reserve(in.size() / 4 * 3 + 3);
    uint32_t acc = 0;
    int bits = 0;
    for (unsigned char c : in)
    {
        const int v = lut[c];
        if (v < 0)
            break;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8)
        {
            bits -= 8;
            out += (char)((acc >> bits) & 0xFF);
        }
    }
    return out;
}

// ---- JSON text out: rapidjson::PrettyWriter with kFormatSingleLineArray and kWriteNanAndInfFlag ------------------
class PrettyWriter
{
  public:
    explicit PrettyWriter(std::string &out) : _o(out)
    {
    }
    void StartObject()
    {
        prefix();
        _levels.push_back(Level{false, 0});
        _o += '{';
    }
    void EndObject()
    {
        const bool empty = _levels.back().count == 0;
        _levels.pop_back();
        if (!empty)
        {
            _o += '\n';
            indent();
        }
        _o += '}';
    }
    void StartArray()
    {
        prefix();
        _levels.push_back(Level{true, 0});
        _o += '[';
    }
    void EndArray()
    {
        _levels.pop_back(); // single-line arrays: no line break before the bracket
        _o += ']';
    }
    void Key(const std::string &s)
    {
        String(s);
    }
    void String(const std::string &s)
    {
        prefix();
        _o += '"';
        for (unsigned char c : s)
        {
            switch (c)
            {
            case '"':
                _o += "\\\"";
                break;
            case '\\':
                _o += "\\\\";
                break;
            case '\b':
                _o += "\\b";
                break;
            case '\f':
                _o += "\\f";
                break;
            case '\n':
                _o += "\\n";
                break;
            case '\r':
                _o += "\\r";
                break;
            case '\t':
                _o += "\\t";
                break;
            default:
                if (c < 0x20)
                {
                    const char hex[] = "0123456789ABCDEF";
                    _o += "\\u00";
                    _o += hex[c >> 4];
                    _o += hex[c & 15];
                }
                else
                    _o += (char)c;
            }
        }
        _o += '"';
    }
    void Int64(int64_t v)
    {
        prefix();
        _o += std::to_string(v);
    }
    void Uint64(uint64_t v)
    {
        prefix();
        _o += std::to_string(v);
    }
    void Bool(bool v)
    {
        prefix();
        _o += v ? "true" : "false";
    }
    void Null()
    {
        prefix();
        _o += "null";
    }
    // Writer::WriteDouble -> internal::dtoa + Prettify (digits, then where the decimal point / exponent go)
    void Double(double d)
    {
        prefix();
        if (std::isnan(d))
        {
            _o += "NaN";
            return;
        }
        if (std::isinf(d))
        {
            _o += d < 0 ? "-Infinity" : "Infinity";
            return;
        }
        if (d == 0)
        {
            _o += std::signbit(d) ? "-0.0" : "0.0";
            return;
        }
        if (d < 0)
        {
            _o += '-';
            d = -d;
        }
        char buf[64];
        const auto r = std::to_chars(buf, buf + sizeof buf - 1, d, std::chars_format::scientific); // d[.ddd]e[+-]XX, shortest
        *r.ptr = '\0';
        std::string digits;
        const char *p = buf;
        for (; p < r.ptr && *p != 'e'; p++)
            if (*p != '.')
                digits += *p;
        const int e10 = std::atoi(p + 1);
        const int length = (int)digits.size(), kk = e10 + 1, k = kk - length;
        if (0 <= k && kk <= 21)
        {
            _o += digits;
            _o.append((size_t)k, '0');
            _o += ".0";
        }
        else if (0 < kk && kk <= 21)
        {
            _o.append(digits, 0, (size_t)kk);
            _o += '.';
            _o.append(digits, (size_t)kk, std::string::npos);
        }
        else if (-6 < kk && kk <= 0)
        {
            _o += "0.";
            _o.append((size_t)(-kk), '0');
            _o += digits;
        }
        else
        {
            _o += digits[0];
            if (length > 1)
            {
                _o += '.';
                _o.append(digits, 1, std::string::npos);
            }
            _o += 'e';
            _o += std::to_string(kk - 1);
        }
    }

  private:
    struct Level
    {
        bool in_array;
        size_t count;
    };
    void indent()
    {
        _o.append(_levels.size() * 4, ' ');
    }
    void prefix() // PrettyWriter::PrettyPrefix
    {
        if (_levels.empty())
            return;
        Level &l = _levels.back();
        if (l.in_array)
        {
            if (l.count > 0)
                _o += ", ";
        }
        else
        {
            if (l.count > 0)
            {
                if (l.count % 2 == 0)
                    _o += ",\n";
                else
                    _o += ": ";
            }
            else
                _o += '\n';
            if (l.count % 2 == 0)
                indent();
        }
        l.count++;
    }
    std::string &_o;
    std::vector<Level> _levels;
};

// ---- JSON text in --------------------------------------------------------------------------------------------------
struct Number
{
    double d = 0;
    bool is_integer = false, negative = false;
    uint64_t u = 0; // magnitude when is_integer
};

class Cursor
{
  public:
    Cursor(const char *b, const char *e) : p(b), end(e)
    {
    }
    const char *p, *end;
    std::string error;

    bool fail(const std::string &what)
    {
        if (error.empty())
            error = what + " at offset " + std::to_string((size_t)(p - _begin()));
        return false;
    }
    void ws()
    {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t'))
            p++;
    }
    bool consume(char c)
    {
        ws();
        if (p < end && *p == c)
        {
            p++;
            return true;
        }
        return false;
    }
    bool expect(char c)
    {
        if (consume(c))
            return true;
        return fail(std::string("expected '") + c + "'");
    }
    char peek()
    {
        ws();
        return p < end ? *p : '\0';
    }
    bool string(std::string &out)
    {
        out.clear();
        if (!expect('"'))
            return false;
        while (p < end && *p != '"')
        {
            if (*p != '\\')
            {
                out += *p++;
                continue;
            }
            if (++p >= end)
                break;
            switch (*p++)
            {
            case '"':
                out += '"';
                break;
            case '\\':
                out += '\\';
                break;
            case '/':
                out += '/';
                break;
            case 'b':
                out += '\b';
                break;
            case 'f':
                out += '\f';
                break;
            case 'n':
                out += '\n';
                break;
            case 'r':
                out += '\r';
                break;
            case 't':
                out += '\t';
                break;
            case 'u': {
                unsigned cp = 0;
                if (!hex4(cp))
                    return fail("bad \\u escape");
                if (cp >= 0xD800 && cp <= 0xDBFF && p + 1 < end && p[0] == '\\' && p[1] == 'u')
                {
                    p += 2;
                    unsigned lo = 0;
                    if (!hex4(lo))
                        return fail("bad \\u escape");
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(cp, out);
                break;
            }
            default:
                return fail("bad escape");
            }
        }
        if (p >= end)
            return fail("unterminated string");
        p++;
        return true;
    }
    bool number(Number &n)
    {
        ws();
        const char *s = p;
        n = Number();
        if (p < end && *p == '-')
        {
            n.negative = true;
            p++;
        }
        if (end - p >= 3 && !std::strncmp(p, "NaN", 3))
        {
            p += 3;
            n.d = NAN;
            return true;
        }
        if (end - p >= 8 && !std::strncmp(p, "Infinity", 8))
        {
            p += 8;
            n.d = n.negative ? -INFINITY : INFINITY;
            return true;
        }
        if (end - p >= 3 && !std::strncmp(p, "Inf", 3))
        {
            p += 3;
            n.d = n.negative ? -INFINITY : INFINITY;
            return true;
        }
        const char *digits = p;
        while (p < end && *p >= '0' && *p <= '9')
            p++;
        if (p == digits)
            return fail("expected a number");
        bool integer = true;
        if (p < end && *p == '.')
        {
            integer = false;
            p++;
            while (p < end && *p >= '0' && *p <= '9')
                p++;
        }
        if (p < end && (*p == 'e' || *p == 'E'))
        {
            integer = false;
            p++;
            if (p < end && (*p == '+' || *p == '-'))
                p++;
            while (p < end && *p >= '0' && *p <= '9')
                p++;
        }
        const std::string tok(s, p);
        n.d = std::strtod(tok.c_str(), nullptr); // correctly rounded: what kParseFullPrecisionFlag asks of rapidjson
        if (integer && p - digits <= 20)
        {
            errno = 0;
            const unsigned long long u = std::strtoull(std::string(digits, p).c_str(), nullptr, 10);
            if (errno == 0)
            {
                n.is_integer = true;
                n.u = u;
            }
        }
        return true;
    }
    bool real(double &d) // GetDouble(): any number
    {
        Number n;
        if (!number(n))
            return false;
        d = n.d;
        return true;
    }
    bool int64(int64_t &v) // GetInt64(): an integer token
    {
        Number n;
        if (!number(n))
            return false;
        if (!n.is_integer)
            return fail("expected an integer");
        v = n.negative ? -(int64_t)n.u : (int64_t)n.u;
        return true;
    }
    bool uint64(uint64_t &v)
    {
        Number n;
        if (!number(n))
            return false;
        if (!n.is_integer || (n.negative && n.u != 0))
            return fail("expected an unsigned integer");
        v = n.u;
        return true;
    }
    bool literal(const char *word)
    {
        ws();
        const size_t l = std::strlen(word);
        if ((size_t)(end - p) >= l && !std::strncmp(p, word, l))
        {
            p += l;
            return true;
        }
        return false;
    }
    bool skip_value()
    {
        const char c = peek();
        std::string s;
        if (c == '{')
        {
            return object([&](const std::string &) { return skip_value(); });
        }
        if (c == '[')
        {
            return array([&]() { return skip_value(); });
        }
        if (c == '"')
            return string(s);
        if (literal("true") || literal("false") || literal("null"))
            return true;
        Number n;
        return number(n);
    }
    // the raw text of the next value
    bool raw_value(std::string &out)
    {
        ws();
        const char *s = p;
        if (!skip_value())
            return false;
        out.assign(s, p);
        return true;
    }
    template <typename F> bool object(F &&member) // member(key) consumes the value
    {
        if (!expect('{'))
            return false;
        if (consume('}'))
            return true;
        std::string key;
        do
        {
            if (!string(key) || !expect(':') || !member(key))
                return false;
        } while (consume(','));
        return expect('}');
    }
    template <typename F> bool array(F &&element)
    {
        if (!expect('['))
            return false;
        if (consume(']'))
            return true;
        do
        {
            if (!element())
                return false;
        } while (consume(','));
        return expect(']');
    }
    template <size_t N> bool reals(double (&out)[N])
    {
        size_t i = 0;
        const bool ok = array([&]() {
            double d;
            if (!real(d))
                return false;
            if (i < N)
                out[i] = d;
            i++;
            return true;
        });
        if (ok && i < N)
            return fail("array too short");
        return ok;
    }
    void set_begin(const char *b)
    {
        _b = b;
    }

  private:
    const char *_b = nullptr;
    const char *_begin() const
    {
        return _b ? _b : p;
    }
    bool hex4(unsigned &cp)
    {
        if (end - p < 4)
            return false;
        cp = 0;
        for (int i = 0; i < 4; i++)
        {
            const char c = *p++;
            cp <<= 4;
            if (c >= '0' && c <= '9')
                cp |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f')
                cp |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                cp |= (unsigned)(c - 'A' + 10);
            else
                return false;
        }
        return true;
    }
    static void utf8(unsigned cp, std::string &out)
    {
        if (cp < 0x80)
            out += (char)cp;
        else if (cp < 0x800)
        {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        }
        else if (cp < 0x10000)
        {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
        else
        {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
};

// re-emit any JSON value through the writer (the metadata object a file carried)
bool copy_value(Cursor &c, PrettyWriter &w)
{
    const char ch = c.peek();
    if (ch == '{')
    {
        w.StartObject();
        const bool ok = c.object([&](const std::string &key) {
            w.Key(key);
            return copy_value(c, w);
        });
        w.EndObject();
        return ok;
    }
    if (ch == '[')
    {
        w.StartArray();
        const bool ok = c.array([&]() { return copy_value(c, w); });
        w.EndArray();
        return ok;
    }
    if (ch == '"')
    {
        std::string s;
        if (!c.string(s))
            return false;
        w.String(s);
        return true;
    }
    if (c.literal("true"))
    {
        w.Bool(true);
        return true;
    }
    if (c.literal("false"))
    {
        w.Bool(false);
        return true;
    }
    if (c.literal("null"))
    {
        w.Null();
        return true;
    }
    Number n;
    if (!c.number(n))
        return false;
    if (n.is_integer)
    {
        if (n.negative)
            w.Int64(-(int64_t)n.u);
        else
            w.Uint64(n.u);
    }
    else
        w.Double(n.d);
    return true;
}

void write_default_metadata(PrettyWriter &w) // image_metadata's defaults (types/image_metadata.hpp:11-58)
{
    w.StartObject();
    w.Key("camera_info");
    w.StartObject();
    w.Key("dimensions");
    w.StartArray();
    w.Uint64(0);
    w.Uint64(0);
    w.EndArray();
    w.Key("focal_length_px");
    w.Double(NAN);
    w.Key("principal");
    w.StartArray();
    w.Double(NAN);
    w.Double(NAN);
    w.EndArray();
    for (const char *k : {"make", "model", "serial_no", "lens_make", "lens_model"})
    {
        w.Key(k);
        w.String("");
    }
    w.EndObject();
    w.Key("capture_info");
    w.StartObject();
    for (const char *k : {"latitude", "longitude", "altitude", "relative_altitude", "roll", "pitch", "yaw", "accuracy_xy", "accuracy_z"})
    {
        w.Key(k);
        w.Double(NAN);
    }
    for (const char *k : {"datum", "timestamp", "datestamp"})
    {
        w.Key(k);
        w.String("");
    }
    w.EndObject();
    w.EndObject();
}

bool parse_id(const std::string &s, size_t &id) // std::strtoull(name, &end, 10)
{
    if (s.empty())
        return false;
    id = (size_t)std::strtoull(s.c_str(), nullptr, 10);
    return true;
}

} // namespace

// ---- MeasurementGraph ------------------------------------------------------------------------------------------------
bool serialize(const MeasurementGraph &graph, std::ostream &out)
{
    // the writer only ever appends, so the text goes to the stream a megabyte at a time: a 1 000-image graph.json is
    // ~2 GB of features and matches and must not be held as one string beside the graph
    std::string text;
    const auto drain = [&](size_t keep) {
        if (text.size() > keep)
        {
            out.write(text.data(), (std::streamsize)text.size());
            text.clear();
        }
    };
    PrettyWriter w(text);
    w.StartObject();
    w.Key("version");
    w.Int64(1);

    w.Key("nodes");
    w.StartObject();
    std::vector<size_t> node_order(graph.size_nodes());
    for (size_t i = 0; i < node_order.size(); i++)
        node_order[i] = i;
    std::sort(node_order.begin(), node_order.end(),
              [&](size_t a, size_t b) { return graph.nodes()[a].id < graph.nodes()[b].id; }); // sorted by id (:227-233)
    for (size_t ni : node_order)
    {
        const MeasurementGraph::Node &node = graph.nodes()[ni];
        const image &img = node.payload;
        drain(1 << 20);
        w.Key(std::to_string(node.id));
        w.StartObject();
        w.Key("path");
        w.String(img.path);
        w.Key("position");
        w.StartArray();
        for (int i = 0; i < 3; i++)
            w.Double(img.position[i]);
        w.EndArray();
        w.Key("orientation");
        w.StartArray();
        for (int i = 0; i < 4; i++)
            w.Double(img.orientation[i]);
        w.EndArray();
        w.Key("thumbnail");
        w.String(img.thumbnail_b64);
        w.Key("model");
        w.StartObject();
        {
            const CameraModel defaults;
            const CameraModel &m = img.model ? *img.model : defaults;
            w.Key("id");
            w.Int64((int64_t)m.id);
            w.Key("dimensions");
            w.StartArray();
            w.Uint64(m.pixels_cols);
            w.Uint64(m.pixels_rows);
            w.EndArray();
            w.Key("focal_length");
            w.Double(m.focal_length_pixels);
            w.Key("principal");
            w.StartArray();
            w.Double(m.principle_point[0]);
            w.Double(m.principle_point[1]);
            w.EndArray();
            w.Key("radial_distortion");
            w.StartArray();
            for (int i = 0; i < 3; i++)
                w.Double(m.radial_distortion[i]);
            w.EndArray();
            w.Key("tangential_distortion");
            w.StartArray();
            for (int i = 0; i < 2; i++)
                w.Double(m.tangential_distortion[i]);
            w.EndArray();
            w.Key("projection");
            w.String("planar");
        }
        w.EndObject();
        w.Key("edges");
        w.StartArray();
        {
            std::vector<size_t> sorted_edges(node.edges);
            std::sort(sorted_edges.begin(), sorted_edges.end());
            for (size_t e : sorted_edges)
                w.String(std::to_string(e));
        }
        w.EndArray();
        w.Key("metadata");
        {
            bool copied = false;
            if (!img.metadata_json.empty())
            {
                Cursor c(img.metadata_json.data(), img.metadata_json.data() + img.metadata_json.size());
                std::string probe;
                PrettyWriter dry(probe);
                Cursor c2 = c;
                if (copy_value(c2, dry)) // well-formed: emit it for real
                    copied = copy_value(c, w);
            }
            if (!copied)
                write_default_metadata(w);
        }
        w.Key("features");
        w.StartArray();
        for (const feature_2d &f : img.features)
        {
            w.StartObject();
            w.Key("location");
            w.StartArray();
            w.Double(f.location[0]);
            w.Double(f.location[1]);
            w.EndArray();
            w.Key("strength");
            w.Double((double)f.strength);
            w.Key("descriptor");
            unsigned char bytes[64];
            std::memcpy(bytes, f.descriptor, 64);
            bytes[60] &= 0x3F; // bits 486.. are not part of the descriptor
            w.String(base64_encode(bytes, (feature_2d::DESCRIPTOR_BITS + 7) >> 3));
            w.EndObject();
        }
        w.EndArray();
        w.Key("num_sparse_features");
        w.Uint64(img.num_sparse_features);
        w.EndObject();
    }
    w.EndObject();

    w.Key("edges");
    w.StartObject();
    std::vector<size_t> edge_order(graph.size_edges());
    for (size_t i = 0; i < edge_order.size(); i++)
        edge_order[i] = i;
    std::sort(edge_order.begin(), edge_order.end(), [&](size_t a, size_t b) { return graph.edges()[a].id < graph.edges()[b].id; });
    for (size_t ei : edge_order)
    {
        const MeasurementGraph::Edge &edge = graph.edges()[ei];
        const camera_relations &r = edge.payload;
        drain(1 << 20);
        w.Key(std::to_string(edge.id));
        w.StartObject();
        w.Key("source");
        w.String(std::to_string(edge.source));
        w.Key("dest");
        w.String(std::to_string(edge.dest));
        w.Key("matches");
        w.StartArray();
        for (const feature_match &m : r.matches)
        {
            w.StartArray();
            w.Int64((int64_t)m.feature_index_1);
            w.Int64((int64_t)m.feature_index_2);
            w.Double(m.distance);
            w.EndArray();
        }
        w.EndArray();
        w.Key("inlier_matches");
        w.StartArray();
        for (const feature_match_denormalized &m : r.inlier_matches)
        {
            w.StartArray();
            w.StartArray();
            w.Double(m.pixel_1[0]);
            w.Double(m.pixel_1[1]);
            w.EndArray();
            w.StartArray();
            w.Double(m.pixel_2[0]);
            w.Double(m.pixel_2[1]);
            w.EndArray();
            w.Int64((int64_t)m.feature_index_1);
            w.Int64((int64_t)m.feature_index_2);
            w.Int64((int64_t)m.match_index);
            w.EndArray();
        }
        w.EndArray();
        w.Key("relation");
        w.StartArray();
        for (int i = 0; i < 9; i++)
            w.Double(r.ransac_relation[i]);
        w.EndArray();
        w.Key("relation_type");
        switch (r.relationType)
        {
        case camera_relations::RelationType::HOMOGRAPHY:
            w.String("homography");
            break;
        case camera_relations::RelationType::FUNDAMENTAL_MATRIX:
            w.String("fundamental_matrix");
            break;
        case camera_relations::RelationType::UNKNOWN:
            w.String("UNKNOWN");
            break;
        }
        w.Key("relative_pose");
        w.StartArray();
        for (const decomposed_pose &pose : r.relative_poses)
        {
            w.StartObject();
            w.Key("score");
            w.Int64(pose.score);
            w.Key("orientation");
            w.StartArray();
            for (int i = 0; i < 4; i++)
                w.Double(pose.orientation[i]);
            w.EndArray();
            w.Key("position");
            w.StartArray();
            for (int i = 0; i < 3; i++)
                w.Double(pose.position[i]);
            w.EndArray();
            w.EndObject();
        }
        w.EndArray();
        w.EndObject();
    }
    w.EndObject();
    w.EndObject();
    drain(0);
    out.flush();
    return (bool)out;
}

namespace
{

bool read_model(Cursor &c, CameraModel &m)
{
    return c.object([&](const std::string &key) {
        if (key == "id")
        {
            int64_t v;
            if (!c.int64(v))
                return false;
            m.id = (size_t)v;
            return true;
        }
        if (key == "dimensions")
        {
            size_t i = 0;
            return c.array([&]() {
                int64_t v;
                if (!c.int64(v))
                    return false;
                if (i == 0)
                    m.pixels_cols = (size_t)v;
                else if (i == 1)
                    m.pixels_rows = (size_t)v;
                i++;
                return true;
            });
        }
        if (key == "focal_length")
            return c.real(m.focal_length_pixels);
        if (key == "principal")
            return c.reals(m.principle_point);
        if (key == "radial_distortion")
            return c.reals(m.radial_distortion);
        if (key == "tangential_distortion")
            return c.reals(m.tangential_distortion);
        return c.skip_value(); // "projection": planar is the only projection on the path
    });
}

bool read_feature(Cursor &c, feature_2d &f)
{
    bool have_descriptor = false;
    const bool ok = c.object([&](const std::string &key) {
        if (key == "location")
            return c.reals(f.location);
        if (key == "strength")
        {
            double d;
            if (!c.real(d))
                return false;
            f.strength = (float)d;
            return true;
        }
        if (key == "descriptor")
        {
            std::string b64;
            if (!c.string(b64))
                return false;
            const std::string bytes = base64_decode(b64);
            if (bytes.size() != (size_t)((feature_2d::DESCRIPTOR_BITS + 7) >> 3))
                return c.fail("descriptor is not 61 bytes");
            unsigned char buf[64] = {0};
            std::memcpy(buf, bytes.data(), bytes.size());
            buf[60] &= 0x3F; // bitset_from_bytes reads bits 0..485
            std::memcpy(f.descriptor, buf, 64);
            have_descriptor = true;
            return true;
        }
        return c.skip_value();
    });
    if (ok && !have_descriptor)
        return c.fail("feature without descriptor");
    return ok;
}

bool read_edge(Cursor &c, camera_relations &r, size_t &source, size_t &dest)
{
    bool have_source = false, have_dest = false;
    const bool ok = c.object([&](const std::string &key) {
        std::string s;
        if (key == "source")
        {
            have_source = true;
            return c.string(s) && parse_id(s, source);
        }
        if (key == "dest")
        {
            have_dest = true;
            return c.string(s) && parse_id(s, dest);
        }
        if (key == "matches")
            return c.array([&]() {
                feature_match fm{0, 0, 0};
                size_t i = 0;
                const bool ok2 = c.array([&]() {
                    if (i < 2)
                    {
                        int64_t v;
                        if (!c.int64(v))
                            return false;
                        (i == 0 ? fm.feature_index_1 : fm.feature_index_2) = (size_t)v;
                    }
                    else if (i == 2)
                    {
                        if (!c.real(fm.distance))
                            return false;
                    }
                    else if (!c.skip_value())
                        return false;
                    i++;
                    return true;
                });
                if (!ok2)
                    return false;
                if (i < 3)
                    return c.fail("match with fewer than 3 entries");
                r.matches.push_back(fm);
                return true;
            });
        if (key == "inlier_matches")
            return c.array([&]() {
                feature_match_denormalized m;
                size_t i = 0;
                const bool ok2 = c.array([&]() {
                    bool good = true;
                    int64_t v = 0;
                    switch (i)
                    {
                    case 0:
                        good = c.reals(m.pixel_1);
                        break;
                    case 1:
                        good = c.reals(m.pixel_2);
                        break;
                    case 2:
                        good = c.int64(v);
                        m.feature_index_1 = (size_t)v;
                        break;
                    case 3:
                        good = c.int64(v);
                        m.feature_index_2 = (size_t)v;
                        break;
                    case 4:
                        good = c.int64(v);
                        m.match_index = (size_t)v;
                        break;
                    default:
                        good = c.skip_value();
                    }
                    i++;
                    return good;
                });
                if (!ok2)
                    return false;
                if (i < 5)
                    return c.fail("inlier match with fewer than 5 entries");
                r.inlier_matches.push_back(m);
                return true;
            });
        if (key == "relation")
            return c.reals(r.ransac_relation);
        if (key == "relation_type")
        {
            if (!c.string(s))
                return false;
            r.relationType = s == "homography"           ? camera_relations::RelationType::HOMOGRAPHY
                             : s == "fundamental_matrix" ? camera_relations::RelationType::FUNDAMENTAL_MATRIX
                                                         : camera_relations::RelationType::UNKNOWN;
            return true;
        }
        if (key == "relative_pose")
        {
            size_t i = 0;
            return c.array([&]() {
                decomposed_pose pose;
                const bool ok2 = c.object([&](const std::string &k2) {
                    if (k2 == "score")
                    {
                        int64_t v;
                        if (!c.int64(v))
                            return false;
                        pose.score = (int)v;
                        return true;
                    }
                    if (k2 == "orientation")
                        return c.reals(pose.orientation);
                    if (k2 == "position")
                        return c.reals(pose.position);
                    return c.skip_value();
                });
                if (ok2 && i < r.relative_poses.size())
                    r.relative_poses[i] = pose;
                i++;
                return ok2;
            });
        }
        return c.skip_value();
    });
    if (ok && !(have_source && have_dest))
        return c.fail("edge without source / dest");
    return ok;
}

} // namespace

bool deserialize(const std::string &json, MeasurementGraph &graph, std::string *error)
{
    Cursor c(json.data(), json.data() + json.size());
    c.set_begin(json.data());
    MeasurementGraph g;
    std::map<size_t, std::shared_ptr<CameraModel>> camera_models; // one shared model per id (:86-110)
    bool version_ok = false, have_nodes = false, have_edges = false;
    auto done = [&](bool ok) {
        if (!ok && error)
            *error = c.error.empty() ? "not a version-1 graph" : c.error;
        return ok;
    };
    if (c.peek() != '{')
        return done(false);
    const bool ok = c.object([&](const std::string &key) {
        if (key == "version")
        {
            Number n;
            if (!c.number(n))
                return false;
            version_ok = n.is_integer && !n.negative && n.u == 1;
            return true;
        }
        if (key == "nodes")
        {
            have_nodes = true;
            return c.object([&](const std::string &id_text) {
                size_t node_id;
                if (!parse_id(id_text, node_id))
                    return c.fail("bad node id");
                image img;
                std::vector<size_t> edge_ids;
                bool have_sparse = false, have_model = false;
                const bool node_ok = c.object([&](const std::string &k) {
                    if (k == "path")
                        return c.string(img.path);
                    if (k == "position")
                        return c.reals(img.position);
                    if (k == "orientation")
                        return c.reals(img.orientation);
                    if (k == "thumbnail")
                        return c.string(img.thumbnail_b64);
                    if (k == "model")
                    {
                        CameraModel m;
                        if (!read_model(c, m))
                            return false;
                        auto it = camera_models.find(m.id);
                        if (it == camera_models.end())
                            it = camera_models.emplace(m.id, std::make_shared<CameraModel>(m)).first;
                        img.model = it->second; // a later copy of a known id is ignored, as in the reference
                        have_model = true;
                        return true;
                    }
                    if (k == "edges")
                        return c.array([&]() {
                            std::string s;
                            size_t e;
                            if (!c.string(s) || !parse_id(s, e))
                                return false;
                            if (std::find(edge_ids.begin(), edge_ids.end(), e) == edge_ids.end()) // a set in the reference
                                edge_ids.push_back(e);
                            return true;
                        });
                    if (k == "metadata")
                        return c.raw_value(img.metadata_json);
                    if (k == "features")
                        return c.array([&]() {
                            feature_2d f;
                            if (!read_feature(c, f))
                                return false;
                            img.features.push_back(f);
                            return true;
                        });
                    if (k == "num_sparse_features")
                    {
                        uint64_t v;
                        if (!c.uint64(v))
                            return false;
                        img.num_sparse_features = (size_t)v;
                        have_sparse = true;
                        return true;
                    }
                    return c.skip_value();
                });
                if (!node_ok)
                    return false;
                if (!have_model)
                    return c.fail("node without camera model");
                if (!have_sparse)
                    img.num_sparse_features = img.features.size(); // files from before the field existed (:199-206)
                if (!g.insertNode(node_id, std::move(img), std::move(edge_ids)))
                    return c.fail("duplicate node id");
                return true;
            });
        }
        if (key == "edges")
        {
            have_edges = true;
            return c.object([&](const std::string &id_text) {
                size_t edge_id, source = 0, dest = 0;
                if (!parse_id(id_text, edge_id))
                    return c.fail("bad edge id");
                camera_relations r;
                if (!read_edge(c, r, source, dest))
                    return false;
                if (!g.insertEdge(edge_id, std::move(r), source, dest))
                    return c.fail("duplicate edge id");
                return true;
            });
        }
        return c.skip_value();
    });
    if (!ok || !version_ok || !have_nodes || !have_edges)
        return done(false);
    c.ws();
    if (c.p != c.end)
    {
        c.fail("text after the document");
        return done(false);
    }
    for (const MeasurementGraph::Edge &e : g.edges())
        if (!g.getNode(e.source) || !g.getNode(e.dest))
        {
            c.error = "edge " + std::to_string(e.id) + " names a node that is not in the file";
            return done(false);
        }
    graph = std::move(g);
    return true;
}

// ---- MeshGraph as PLY ------------------------------------------------------------------------------------------------
bool serialize(const MeshGraph &graph, std::ostream &out)
{
    const char nl = '\n';
    out << "ply" << nl << "format ascii 1.0" << nl << "comment exported from OpenCalibration" << nl;
    out << "element vertex " << graph.size_nodes() << nl;
    out << "property double x" << nl << "property double y" << nl << "property double z" << nl << "property int nodeIndex" << nl;

    // every edge contributes the triangle(s) it borders; a triangle's corners sorted by id, the first two swapped when
    // that order is anticlockwise (geometry/utils.hpp:10-14: cross z < 0), each triangle once (:55-85)
    std::set<std::array<size_t, 3>> faces;
    for (const MeshEdge &e : graph.edges)
    {
        auto add_face = [&](size_t opposite) {
            std::array<size_t, 3> face{e.source, e.dest, opposite == MeshEdge::NONE ? 0 : opposite};
            std::sort(face.begin(), face.end());
            const double *p0 = graph.nodes[face[0]].location, *p1 = graph.nodes[face[1]].location, *p2 = graph.nodes[face[2]].location;
            const double cross_z = (p1[0] - p0[0]) * (p2[1] - p0[1]) - (p1[1] - p0[1]) * (p2[0] - p0[0]);
            if (cross_z < 0)
                std::swap(face[0], face[1]);
            faces.insert(face);
        };
        add_face(e.triangleOppositeNodes[0]);
        if (!e.border)
            add_face(e.triangleOppositeNodes[1]);
    }
    out << "element face " << faces.size() << nl;
    out << "property list uchar int vertex_index" << nl;
    out << "element edge " << graph.size_edges() << nl;
    out << "property int vertex1" << nl << "property int vertex2" << nl << "property int edgeIndex" << nl << "property uchar border" << nl
        << "property int oppositeCorner1" << nl << "property int oppositeCorner2" << nl << "end_header" << nl;
    for (size_t i = 0; i < graph.nodes.size(); i++)
    {
        const double *loc = graph.nodes[i].location;
        out << loc[0] << " " << loc[1] << " " << loc[2] << " " << i << nl;
    }
    for (const auto &face : faces) // std::set order = the reference's std::sort of the faces
        out << "3 " << face[0] << " " << face[1] << " " << face[2] << nl;
    for (size_t i = 0; i < graph.edges.size(); i++)
    {
        const MeshEdge &e = graph.edges[i];
        auto corner = [](size_t c) { return c == MeshEdge::NONE ? (size_t)0 : c; }; // an unset corner is 0 in the reference
        out << e.source << " " << e.dest << " " << i << " " << e.border << " " << corner(e.triangleOppositeNodes[0]) << " "
            << corner(e.triangleOppositeNodes[1]) << nl;
    }
    return (bool)out;
}

bool deserialize(std::istream &ply, MeshGraph &graph)
{
    graph = MeshGraph();
    std::string line;
    auto expect_line = [&](const char *x) { return std::getline(ply, line) && line == x; };
    auto split = [](const std::string &s) {
        std::vector<std::string> words;
        std::istringstream iss(s);
        std::string item;
        while (std::getline(iss, item, ' '))
            words.push_back(item);
        return words;
    };
    auto count_after = [&](const char *prefix, size_t &count) {
        if (!std::getline(ply, line) || line.compare(0, std::strlen(prefix), prefix) != 0)
            return false;
        try
        {
            count = (size_t)std::stoll(line.substr(std::strlen(prefix)));
        }
        catch (...)
        {
            return false;
        }
        return true;
    };
    if (!expect_line("ply") || !expect_line("format ascii 1.0") || !expect_line("comment exported from OpenCalibration"))
        return false;
    size_t num_nodes = 0, num_faces = 0, num_edges = 0;
    if (!count_after("element vertex ", num_nodes))
        return false;
    if (!expect_line("property double x") || !expect_line("property double y") || !expect_line("property double z") ||
        !expect_line("property int nodeIndex"))
        return false;
    if (!count_after("element face ", num_faces) || !expect_line("property list uchar int vertex_index"))
        return false;
    if (!count_after("element edge ", num_edges))
        return false;
    if (!expect_line("property int vertex1") || !expect_line("property int vertex2") || !expect_line("property int edgeIndex") ||
        !expect_line("property uchar border") || !expect_line("property int oppositeCorner1") ||
        !expect_line("property int oppositeCorner2") || !expect_line("end_header"))
        return false;
    try
    {
        std::unordered_map<size_t, size_t> index_of_id; // the file's node ids -> insertion index
        for (size_t i = 0; i < num_nodes; i++)
        {
            if (!std::getline(ply, line))
                return false;
            const auto words = split(line);
            if (words.size() != 4)
                return false;
            graph.addNode(std::stod(words[0]), std::stod(words[1]), std::stod(words[2]));
            if (!index_of_id.emplace((size_t)std::stoull(words[3]), i).second)
                return false;
        }
        for (size_t i = 0; i < num_faces; i++)
        {
            if (!std::getline(ply, line) || split(line).size() != 4)
                return false;
        }
        for (size_t i = 0; i < num_edges; i++)
        {
            if (!std::getline(ply, line))
                return false;
            const auto words = split(line);
            if (words.size() != 6)
                return false;
            const size_t s = (size_t)std::stoull(words[0]), d = (size_t)std::stoull(words[1]);
            if (s >= num_nodes || d >= num_nodes)
                return false;
            MeshEdge e;
            e.border = (bool)std::stoi(words[3]);
            for (int k = 0; k < 2; k++)
            {
                if (k == 1 && e.border)
                    continue; // unset in the reference (written as 0), unused
                const auto it = index_of_id.find((size_t)std::stoull(words[4 + k]));
                if (it == index_of_id.end())
                    return false;
                e.triangleOppositeNodes[k] = it->second;
            }
            graph.addEdge(e, s, d); // edge ids become file order (std::stoull(words[2]) is the writer's id)
        }
    }
    catch (...) // std::stod / std::stoull on a malformed word: the reference would throw out of deserialize
    {
        return false;
    }
    if (std::getline(ply, line))
        return false;
    return true;
}

// ---- checkpoint directory ----------------------------------------------------------------------------------------
namespace
{
const char *const PIPELINE_STATES[] = {"INITIAL_PROCESSING", "INITIAL_GLOBAL_RELAX", "CAMERA_PARAMETER_RELAX", "FINAL_GLOBAL_RELAX",
                                       "MESH_REFINEMENT",    "GENERATE_THUMBNAIL",   "DENSIFY_MESH",           "DENSE_MESH_RELAX",
                                       "GENERATE_LAYERS",    "COLOR_BALANCE",        "BLEND_LAYERS",           "COMPLETE"};
bool known_state(const std::string &s)
{
    for (const char *k : PIPELINE_STATES)
        if (s == k)
            return true;
    return false;
}
bool set_error(std::string *error, const std::string &what)
{
    if (error)
        *error = what;
    return false;
}
} // namespace

bool saveCheckpoint(const CheckpointData &data, const std::string &checkpoint_dir, std::string *error)
{
    namespace fs = std::filesystem;
    const fs::path dir(checkpoint_dir);
    std::error_code ec;
    fs::create_directories(dir, ec);
    if (ec)
        return set_error(error, "Failed to create checkpoint directory: " + ec.message());
    {
        // metadata.json: the default PrettyWriter (arrays are not on the path here)
        std::string text;
        PrettyWriter w(text);
        w.StartObject();
        w.Key("version");
        w.Int64(1);
        w.Key("state");
        w.String(data.state);
        w.Key("state_run_count");
        w.Uint64(data.state_run_count);
        w.Key("origin_latitude");
        w.Double(data.origin_latitude);
        w.Key("origin_longitude");
        w.Double(data.origin_longitude);
        w.Key("surface_count");
        w.Uint64(data.surfaces.size());
        w.EndObject();
        std::ofstream out(dir / "metadata.json");
        if (!out.is_open())
            return set_error(error, "Failed to open metadata.json for writing");
        out << text;
    }
    {
        std::ofstream out(dir / "graph.json");
        if (!out.is_open())
            return set_error(error, "Failed to open graph.json for writing");
        if (!serialize(data.graph, out))
            return set_error(error, "Failed to serialize graph");
    }
    for (size_t i = 0; i < data.surfaces.size(); i++)
    {
        const surface_model &surface = data.surfaces[i];
        if (surface.mesh.size_nodes() > 0)
        {
            const std::string name = "surface_" + std::to_string(i) + ".ply";
            std::ofstream out(dir / name);
            if (!out.is_open())
                return set_error(error, "Failed to open " + name + " for writing");
            if (!serialize(surface.mesh, out))
                return set_error(error, "Failed to serialize mesh " + std::to_string(i));
        }
        for (size_t j = 0; j < surface.cloud.size(); j++)
        {
            const std::string name = "pointcloud_" + std::to_string(i) + "_" + std::to_string(j) + ".xyz";
            std::ofstream out(dir / name);
            if (!out.is_open())
                return set_error(error, "Failed to open " + name + " for writing");
            for (const auto &p : surface.cloud[j]) // `ostream << double`: 6 significant digits, as the reference (:119-122)
                out << p[0] << "," << p[1] << "," << p[2] << "\n";
        }
        std::ofstream count_out(dir / ("surface_" + std::to_string(i) + "_cloudcount.txt"));
        if (count_out.is_open())
            count_out << surface.cloud.size();
    }
    return true;
}

bool loadCheckpoint(const std::string &checkpoint_dir, CheckpointData &data, std::string *error)
{
    namespace fs = std::filesystem;
    const fs::path dir(checkpoint_dir);
    if (!fs::exists(dir))
        return set_error(error, "Checkpoint directory does not exist: " + checkpoint_dir);
    auto slurp = [](const fs::path &p, std::string &text) {
        std::ifstream in(p);
        if (!in.is_open())
            return false;
        text.assign((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        return true;
    };
    size_t surface_count = 0;
    {
        std::string text;
        if (!slurp(dir / "metadata.json", text))
            return set_error(error, "Failed to open metadata.json for reading");
        Cursor c(text.data(), text.data() + text.size());
        c.set_begin(text.data());
        bool version_ok = false;
        if (c.peek() != '{')
            return set_error(error, "Failed to parse metadata.json");
        const bool ok = c.object([&](const std::string &key) {
            if (key == "version")
            {
                Number n;
                if (!c.number(n))
                    return false;
                version_ok = n.is_integer && !n.negative && n.u == 1;
                return true;
            }
            if (key == "state")
            {
                std::string s;
                if (!c.string(s))
                    return false;
                data.state = known_state(s) ? s : "INITIAL_PROCESSING"; // value_or(INITIAL_PROCESSING) (:86-87)
                return true;
            }
            if (key == "state_run_count")
                return c.uint64(data.state_run_count);
            if (key == "origin_latitude")
                return c.real(data.origin_latitude);
            if (key == "origin_longitude")
                return c.real(data.origin_longitude);
            if (key == "surface_count")
            {
                uint64_t v;
                if (!c.uint64(v))
                    return false;
                surface_count = (size_t)v;
                return true;
            }
            return c.skip_value();
        });
        c.ws();
        if (!ok || c.p != c.end)
            return set_error(error, "Failed to parse metadata.json");
        if (!version_ok)
            return set_error(error, "Unsupported checkpoint version");
    }
    {
        std::string text;
        if (!slurp(dir / "graph.json", text))
            return set_error(error, "Failed to open graph.json for reading");
        std::string why;
        if (!deserialize(text, data.graph, &why))
            return set_error(error, "Failed to deserialize graph: " + why);
    }
    data.surfaces.clear();
    data.surfaces.resize(surface_count);
    for (size_t i = 0; i < surface_count; i++)
    {
        surface_model &surface = data.surfaces[i];
        const fs::path mesh_path = dir / ("surface_" + std::to_string(i) + ".ply");
        if (fs::exists(mesh_path))
        {
            std::ifstream in(mesh_path);
            if (in.is_open() && !deserialize(in, surface.mesh))
                surface.mesh = MeshGraph(); // the reference warns and goes on
        }
        size_t cloud_count = 0;
        {
            std::ifstream in(dir / ("surface_" + std::to_string(i) + "_cloudcount.txt"));
            if (in.is_open())
                in >> cloud_count;
        }
        surface.cloud.resize(cloud_count);
        for (size_t j = 0; j < cloud_count; j++)
        {
            std::ifstream in(dir / ("pointcloud_" + std::to_string(i) + "_" + std::to_string(j) + ".xyz"));
            if (!in.is_open())
                continue;
            std::string line;
            while (std::getline(in, line))
            {
                if (line.empty())
                    continue;
                const size_t p1 = line.find(','), p2 = p1 == std::string::npos ? p1 : line.find(',', p1 + 1);
                if (p1 == std::string::npos || p2 == std::string::npos)
                    continue;
                try
                {
                    surface.cloud[j].push_back({std::stod(line.substr(0, p1)), std::stod(line.substr(p1 + 1, p2 - p1 - 1)),
                                                std::stod(line.substr(p2 + 1))});
                }
                catch (...)
                {
                    return set_error(error, "malformed point in pointcloud_" + std::to_string(i) + "_" + std::to_string(j) + ".xyz");
                }
            }
        }
    }
    return true;
}

bool validateCheckpoint(const std::string &checkpoint_dir)
{
    namespace fs = std::filesystem;
    const fs::path dir(checkpoint_dir);
    return fs::exists(dir) && fs::exists(dir / "metadata.json") && fs::exists(dir / "graph.json");
}

} // namespace opencalibration_amd
