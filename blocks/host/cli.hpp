// cli.hpp -- the command-line grammar of the reference blocks, rebuilt as a small
// table-driven parser (the reference uses its `opt` library, libs/opt/opt.cpp:104-154,
// 213-234, 251-436 and libs/opt/opt.hpp:564-591, 666-685; behaviour reproduced,
// implementation new):
//   * any argument equal to `help` prints the usage line on stderr and fails;
//   * `key=value` sets the option named key (positional or keyword); unknown keys
//     are silently ignored; an empty value leaves the option unset;
//   * a bare argument naming a bool option sets it to true;
//   * any other bare argument fills the first positional option not yet set
//     (extra ones are ignored);
//   * values are cut at the first whitespace; enum values are case-insensitive;
//   * unsigned integers take C prefixes (0x.., 0..) or a decimal fixed-point value
//     with an SI suffix k K M G T P E; floats take z a f p n u m k M G T P E;
//   * sets are comma-separated lists.
// A failed parse prints `error: <key>: <what>` and the block exits with status 255.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <set>
#include <string>
#include <utility>
#include <vector>

namespace cli {

using Choices = std::vector<std::pair<std::string, int>>;

class Option {
public:
    enum Kind { Uint, Float, Bool, Enum, UintSet, EnumSet };

    static Option uint(std::string key, std::string placeholder, uint64_t dflt = 0, bool required = false);
    static Option real(std::string key, std::string placeholder, float dflt = 0.0f, bool required = false);
    static Option boolean(std::string key, bool dflt = false);
    static Option choice(std::string key, const Choices& choices, int dflt = 0, bool required = false);
    static Option uint_set(std::string key, std::string placeholder);
    static Option choice_set(std::string key, const Choices& choices);

    const std::string& key() const { return key_; }
    bool is_set() const { return set_; }
    bool required() const { return required_; }
    Kind kind() const { return kind_; }
    std::string placeholder() const;

    uint64_t as_uint() const { return u_; }
    float as_float() const { return f_; }
    bool as_bool() const { return b_; }
    int as_choice() const { return e_; }
    const std::set<uint64_t>& as_uint_set() const { return uset_; }
    const std::set<int>& as_choice_set() const { return eset_; }

    // Parses one (already trimmed) value; prints `error: key: ...` on failure.
    bool assign(const std::string& value, std::ostream& err);

private:
    Option() = default;
    bool parse_uint(const std::string& s, uint64_t& out, std::ostream& err);
    bool parse_float(const std::string& s, float& out, std::ostream& err);
    bool parse_choice(const std::string& s, int& out, std::ostream& err);
    std::ostream& complain(std::ostream& err) const;

    Kind kind_ = Uint;
    std::string key_, placeholder_;
    bool required_ = false, set_ = false;
    uint64_t u_ = 0;
    float f_ = 0.0f;
    bool b_ = false;
    int e_ = 0;
    std::set<uint64_t> uset_;
    std::set<int> eset_;
    Choices choices_;
};

// Returns false when the block must exit with status 255 (help or bad value).
bool parse(const std::vector<Option*>& positional, const std::vector<Option*>& keyword, int argc,
           const char* const* argv, std::ostream& err);
void usage(const char* program, const std::vector<Option*>& positional, const std::vector<Option*>& keyword,
           std::ostream& out);

// 0 <= id <= 65535 (include/sdr/options.hpp:79-83 of the reference)
inline bool valid_stream_id(uint64_t id) { return id <= 65535; }

} // namespace cli
