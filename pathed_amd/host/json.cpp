#include "json.h"

#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>

namespace pathed {

class JsonParser {
public:
    explicit JsonParser(const std::string &text) : m_text(text), m_pos(0) {}

    Json parseDocument()
    {
        Json value = parseValue();
        skipWhitespace();
        if (m_pos != m_text.size()) { fail("trailing characters"); }
        return value;
    }

private:
    const std::string &m_text;
    size_t m_pos;
    int m_depth = 0;
    static const int kMaxDepth = 256;   // nesting of arrays / objects: the parser recurses

    struct Nested {
        JsonParser &parser;
        explicit Nested(JsonParser &p) : parser(p)
        {
            if (++parser.m_depth > kMaxDepth) { parser.fail("nesting deeper than 256 levels"); }
        }
        ~Nested() { parser.m_depth--; }
    };

    [[noreturn]] void fail(const std::string &what) const
    {
        throw JsonError("json: " + what + " at offset " + std::to_string(m_pos));
    }

    void skipWhitespace()
    {
        while (m_pos < m_text.size()) {
            char c = m_text[m_pos];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') { m_pos++; }
            else { break; }
        }
    }

    char peek()
    {
        skipWhitespace();
        if (m_pos >= m_text.size()) { fail("unexpected end"); }
        return m_text[m_pos];
    }

    void expect(char c)
    {
        if (peek() != c) { fail(std::string("expected '") + c + "'"); }
        m_pos++;
    }

    Json parseValue()
    {
        char c = peek();
        if (c == '{') { Nested level(*this); return parseObject(); }
        if (c == '[') { Nested level(*this); return parseArray(); }
        if (c == '"') {
            Json j;
            j.m_type = Json::Type::String;
            j.m_string = parseString();
            return j;
        }
        if (c == 't' || c == 'f') { return parseBool(); }
        if (c == 'n') {
            if (m_text.compare(m_pos, 4, "null") != 0) { fail("bad literal"); }
            m_pos += 4;
            return Json();
        }
        return parseNumber();
    }

    Json parseBool()
    {
        Json j;
        j.m_type = Json::Type::Bool;
        if (m_text.compare(m_pos, 4, "true") == 0) {
            j.m_bool = true;
            m_pos += 4;
        } else if (m_text.compare(m_pos, 5, "false") == 0) {
            j.m_bool = false;
            m_pos += 5;
        } else {
            fail("bad literal");
        }
        return j;
    }

    Json parseNumber()
    {
        size_t start = m_pos;
        bool isInteger = true;
        if (m_pos < m_text.size() && (m_text[m_pos] == '-' || m_text[m_pos] == '+')) { m_pos++; }
        while (m_pos < m_text.size()) {
            char c = m_text[m_pos];
            if (std::isdigit((unsigned char)c)) { m_pos++; }
            else if (c == '.' || c == 'e' || c == 'E' || c == '+' || c == '-') {
                isInteger = false;
                m_pos++;
            } else { break; }
        }
        if (start == m_pos) { fail("bad number"); }
        Json j;
        j.m_type = Json::Type::Number;
        j.m_number = std::strtod(m_text.substr(start, m_pos - start).c_str(), nullptr);
        j.m_isInteger = isInteger;
        return j;
    }

    std::string parseString()
    {
        expect('"');
        std::string out;
        while (true) {
            if (m_pos >= m_text.size()) { fail("unterminated string"); }
            char c = m_text[m_pos++];
            if (c == '"') { break; }
            if (c == '\\') {
                if (m_pos >= m_text.size()) { fail("bad escape"); }
                char e = m_text[m_pos++];
                switch (e) {
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case '/': out += '/'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'n': out += '\n'; break;
                case 'r': out += '\r'; break;
                case 't': out += '\t'; break;
                case 'u': {
                    if (m_pos + 4 > m_text.size()) { fail("bad \\u escape"); }
                    unsigned code = (unsigned)std::strtoul(m_text.substr(m_pos, 4).c_str(), nullptr, 16);
                    m_pos += 4;
                    if (code < 0x80) { out += (char)code; }
                    else if (code < 0x800) {
                        out += (char)(0xC0 | (code >> 6));
                        out += (char)(0x80 | (code & 0x3F));
                    } else {
                        out += (char)(0xE0 | (code >> 12));
                        out += (char)(0x80 | ((code >> 6) & 0x3F));
                        out += (char)(0x80 | (code & 0x3F));
                    }
                    break;
                }
                default: fail("bad escape");
                }
            } else {
                out += c;
            }
        }
        return out;
    }

    Json parseArray()
    {
        expect('[');
        Json j;
        j.m_type = Json::Type::Array;
        if (peek() == ']') { m_pos++; return j; }
        while (true) {
            j.m_array.push_back(parseValue());
            char c = peek();
            m_pos++;
            if (c == ']') { break; }
            if (c != ',') { fail("expected ',' or ']'"); }
        }
        return j;
    }

    Json parseObject()
    {
        expect('{');
        Json j;
        j.m_type = Json::Type::Object;
        if (peek() == '}') { m_pos++; return j; }
        while (true) {
            if (peek() != '"') { fail("expected key"); }
            std::string key = parseString();
            expect(':');
            Json value = parseValue();
            bool replaced = false;
            for (auto &item : j.m_object) {
                if (item.first == key) { item.second = value; replaced = true; }
            }
            if (!replaced) { j.m_object.emplace_back(key, value); }
            char c = peek();
            m_pos++;
            if (c == '}') { break; }
            if (c != ',') { fail("expected ',' or '}'"); }
        }
        return j;
    }
};

Json Json::parse(const std::string &text)
{
    JsonParser parser(text);
    return parser.parseDocument();
}

Json Json::parseFile(const std::string &path)
{
    std::ifstream file(path);
    if (!file) { throw JsonError("json: cannot open " + path); }
    std::stringstream buffer;
    buffer << file.rdbuf();
    return parse(buffer.str());
}

static const Json &nullJson()
{
    static const Json instance;
    return instance;
}

const Json &Json::operator[](const std::string &key) const
{
    if (m_type != Type::Object) { return nullJson(); }
    for (const auto &item : m_object) {
        if (item.first == key) { return item.second; }
    }
    return nullJson();
}

const Json &Json::operator[](size_t index) const
{
    if (m_type != Type::Array || index >= m_array.size()) { return nullJson(); }
    return m_array[index];
}

bool Json::has(const std::string &key) const
{
    if (m_type != Type::Object) { return false; }
    for (const auto &item : m_object) {
        if (item.first == key) { return true; }
    }
    return false;
}

size_t Json::size() const
{
    if (m_type == Type::Array) { return m_array.size(); }
    if (m_type == Type::Object) { return m_object.size(); }
    return 0;
}

bool Json::asBool() const
{
    if (m_type != Type::Bool) { throw JsonError("json: value is not a boolean"); }
    return m_bool;
}

double Json::asNumber() const
{
    if (m_type != Type::Number) { throw JsonError("json: value is not a number"); }
    return m_number;
}

int Json::asInt() const
{
    return (int)asNumber();
}

const std::string &Json::asString() const
{
    if (m_type != Type::String) { throw JsonError("json: value is not a string"); }
    return m_string;
}

Json Json::makeObject()
{
    Json j;
    j.m_type = Type::Object;
    return j;
}

Json Json::makeNumber(double v)
{
    Json j;
    j.m_type = Type::Number;
    j.m_number = v;
    j.m_isInteger = (std::floor(v) == v && std::fabs(v) < 1e15);
    return j;
}

Json Json::makeString(const std::string &s)
{
    Json j;
    j.m_type = Type::String;
    j.m_string = s;
    return j;
}

Json Json::makeBool(bool b)
{
    Json j;
    j.m_type = Type::Bool;
    j.m_bool = b;
    return j;
}

void Json::set(const std::string &key, const Json &value)
{
    if (m_type != Type::Object) { throw JsonError("json: set on non-object"); }
    for (auto &item : m_object) {
        if (item.first == key) { item.second = value; return; }
    }
    m_object.emplace_back(key, value);
}

static void escapeTo(std::string &out, const std::string &s)
{
    out += '"';
    for (char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default: out += c;
        }
    }
    out += '"';
}

void Json::dumpTo(std::string &out, int indent, int depth) const
{
    const std::string pad((size_t)indent * (depth + 1), ' ');
    const std::string closePad((size_t)indent * depth, ' ');
    switch (m_type) {
    case Type::Null: out += "null"; break;
    case Type::Bool: out += m_bool ? "true" : "false"; break;
    case Type::Number: {
        char buffer[64];
        if (m_isInteger) { snprintf(buffer, sizeof buffer, "%lld", (long long)m_number); }
        else { snprintf(buffer, sizeof buffer, "%.17g", m_number); }
        out += buffer;
        break;
    }
    case Type::String: escapeTo(out, m_string); break;
    case Type::Array:
        if (m_array.empty()) { out += "[]"; break; }
        out += "[\n";
        for (size_t i = 0; i < m_array.size(); i++) {
            out += pad;
            m_array[i].dumpTo(out, indent, depth + 1);
            out += (i + 1 < m_array.size()) ? ",\n" : "\n";
        }
        out += closePad + "]";
        break;
    case Type::Object:
        if (m_object.empty()) { out += "{}"; break; }
        out += "{\n";
        for (size_t i = 0; i < m_object.size(); i++) {
            out += pad;
            escapeTo(out, m_object[i].first);
            out += ": ";
            m_object[i].second.dumpTo(out, indent, depth + 1);
            out += (i + 1 < m_object.size()) ? ",\n" : "\n";
        }
        out += closePad + "}";
        break;
    }
}

std::string Json::dump(int indent) const
{
    std::string out;
    dumpTo(out, indent, 0);
    return out;
}

}  // namespace pathed
