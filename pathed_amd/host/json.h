// Minimal JSON DOM reader/writer for job.json and scene JSON.
// The reference reads both with nlohmann::json (vendor/json.hpp); only the
// subset those files use is implemented: objects, arrays, strings, numbers,
// booleans, null.  Key order of objects is preserved for report.json.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace pathed {

class Json {
public:
    enum class Type { Null, Bool, Number, String, Array, Object };

    Json() : m_type(Type::Null) {}
    static Json parse(const std::string &text);
    static Json parseFile(const std::string &path);

    Type type() const { return m_type; }
    bool isNull() const { return m_type == Type::Null; }
    bool isBool() const { return m_type == Type::Bool; }
    bool isNumber() const { return m_type == Type::Number; }
    bool isString() const { return m_type == Type::String; }
    bool isArray() const { return m_type == Type::Array; }
    bool isObject() const { return m_type == Type::Object; }

    // object access; a missing key yields a Null value (nlohmann's operator[] on a
    // mutable object inserts null, which the reference relies on for optional keys)
    const Json &operator[](const std::string &key) const;
    const Json &operator[](size_t index) const;
    bool has(const std::string &key) const;
    size_t size() const;
    const std::vector<std::pair<std::string, Json>> &items() const { return m_object; }
    const std::vector<Json> &elements() const { return m_array; }

    bool asBool() const;
    double asNumber() const;
    int asInt() const;
    const std::string &asString() const;

    std::string dump(int indent = 4) const;

    // construction helpers (used when writing report.json / metrics)
    static Json makeObject();
    static Json makeNumber(double v);
    static Json makeString(const std::string &s);
    static Json makeBool(bool b);
    void set(const std::string &key, const Json &value);

private:
    Type m_type;
    bool m_bool = false;
    double m_number = 0.0;
    bool m_isInteger = false;
    std::string m_string;
    std::vector<Json> m_array;
    std::vector<std::pair<std::string, Json>> m_object;

    void dumpTo(std::string &out, int indent, int depth) const;
    friend class JsonParser;
};

struct JsonError : std::runtime_error {
    explicit JsonError(const std::string &what) : std::runtime_error(what) {}
};

}  // namespace pathed
