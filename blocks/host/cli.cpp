#include "cli.hpp"

#include <algorithm>
#include <cctype>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <ostream>

namespace cli {

namespace {

const char kSpace[] = " \t\f\r\n";

// first whitespace-delimited token of s
std::string first_token(const std::string& s) {
    size_t b = s.find_first_not_of(kSpace);
    if (b == std::string::npos) return std::string();
    size_t e = s.find_first_of(kSpace, b);
    return s.substr(b, e == std::string::npos ? std::string::npos : e - b);
}

std::string lower(std::string s) {
    for (char& c : s) c = static_cast<char>(::tolower(static_cast<unsigned char>(c)));
    return s;
}

int si_exponent(char c) {
    switch (c) {
        case 'k': case 'K': return 3;
        case 'M': return 6;
        case 'G': return 9;
        case 'T': return 12;
        case 'P': return 15;
        case 'E': return 18;
        default: return -1;
    }
}

bool si_float_factor(char c, float& f) {
    switch (c) {
        case 'z': f = 1e-21f; return true;
        case 'a': f = 1e-18f; return true;
        case 'f': f = 1e-15f; return true;
        case 'p': f = 1e-12f; return true;
        case 'n': f = 1e-9f; return true;
        case 'u': f = 1e-6f; return true;
        case 'm': f = 1e-3f; return true;
        case 'k': f = 1e3f; return true;
        case 'M': f = 1e6f; return true;
        case 'G': f = 1e9f; return true;
        case 'T': f = 1e12f; return true;
        case 'P': f = 1e15f; return true;
        case 'E': f = 1e18f; return true;
        default: return false;
    }
}

uint64_t pow10u(int e) {
    uint64_t v = 1;
    while (e-- > 0) v *= 10;
    return v;
}

std::string join_choices(const Choices& c) {
    std::string s = "(";
    for (size_t i = 0; i < c.size(); ++i) {
        if (i) s += "|";
        s += c[i].first;
    }
    return s + ")";
}

} // namespace

Option Option::uint(std::string key, std::string placeholder, uint64_t dflt, bool required) {
    Option o;
    o.kind_ = Uint; o.key_ = std::move(key); o.placeholder_ = std::move(placeholder);
    o.u_ = dflt; o.required_ = required;
    return o;
}
Option Option::real(std::string key, std::string placeholder, float dflt, bool required) {
    Option o;
    o.kind_ = Float; o.key_ = std::move(key); o.placeholder_ = std::move(placeholder);
    o.f_ = dflt; o.required_ = required;
    return o;
}
Option Option::boolean(std::string key, bool dflt) {
    Option o;
    o.kind_ = Bool; o.key_ = std::move(key); o.placeholder_ = "(true|1|false|0)"; o.b_ = dflt;
    return o;
}
Option Option::choice(std::string key, const Choices& choices, int dflt, bool required) {
    Option o;
    o.kind_ = Enum; o.key_ = std::move(key); o.choices_ = choices; o.e_ = dflt; o.required_ = required;
    return o;
}
Option Option::uint_set(std::string key, std::string placeholder) {
    Option o;
    o.kind_ = UintSet; o.key_ = std::move(key); o.placeholder_ = std::move(placeholder);
    return o;
}
Option Option::choice_set(std::string key, const Choices& choices) {
    Option o;
    o.kind_ = EnumSet; o.key_ = std::move(key); o.choices_ = choices;
    return o;
}

std::string Option::placeholder() const {
    if (kind_ == Enum) return join_choices(choices_);
    if (kind_ == EnumSet) return join_choices(choices_) + ",...";
    return placeholder_;
}

std::ostream& Option::complain(std::ostream& err) const { return err << "error: " << key_ << ": "; }

bool Option::parse_uint(const std::string& s, uint64_t& out, std::ostream& err) {
    const char* what = "unsigned integer value expected";
    if (s[0] == '-') { complain(err) << what << std::endl; return false; }
    errno = 0;
    char* end = nullptr;
    unsigned long long v = std::strtoull(s.c_str(), &end, 0);
    if (end == s.c_str()) { complain(err) << what << std::endl; return false; }
    if (errno == ERANGE) {
        // the reference reports the overflow but keeps going with the old value
        complain(err) << "option value out of range" << std::endl;
        return true;
    }
    std::string rest(end);
    if (!rest.empty()) {
        // fixed point and SI suffix: decimal values only
        if (s[0] == '0') { complain(err) << what << std::endl; return false; }
        uint64_t frac = 0;
        int frac_digits = 0;
        if (rest[0] == '.') {
            size_t d = 1;
            while (d < rest.size() && std::isdigit(static_cast<unsigned char>(rest[d]))) ++d;
            if (d == 1) { complain(err) << "digits expected after point" << std::endl; return false; }
            if (d == rest.size()) { complain(err) << what << std::endl; return false; } // fraction without unit
            frac_digits = static_cast<int>(d - 1);
            if (frac_digits > 19) { complain(err) << "option value out of range" << std::endl; return true; }
            frac = std::strtoull(rest.substr(1, d - 1).c_str(), nullptr, 10);
            rest.erase(0, d);
        }
        int exp = si_exponent(rest[0]);
        if (exp < 0 || rest.size() > 1) { complain(err) << "invalid unit '" << rest << '\'' << std::endl; return false; }
        if (frac_digits > exp) { complain(err) << what << std::endl; return false; }
        v = v * pow10u(exp) + frac * pow10u(exp - frac_digits);
    }
    out = v;
    return true;
}

bool Option::parse_float(const std::string& s, float& out, std::ostream& err) {
    errno = 0;
    char* end = nullptr;
    float v = std::strtof(s.c_str(), &end);
    if (end == s.c_str()) { complain(err) << "floating-point value expected" << std::endl; return false; }
    if (errno == ERANGE) { complain(err) << "option value out of range" << std::endl; return true; }
    if (*end) {
        float factor;
        if (!si_float_factor(*end, factor) || end[1]) {
            complain(err) << "invalid unit '" << end << '\'' << std::endl;
            return false;
        }
        v *= factor;
    }
    out = v;
    return true;
}

bool Option::parse_choice(const std::string& s, int& out, std::ostream& err) {
    const std::string lc = lower(s);
    for (const auto& c : choices_)
        if (c.first == lc) { out = c.second; return true; }
    complain(err) << "invalid value '" << s << "'" << std::endl;
    return false;
}

bool Option::assign(const std::string& value, std::ostream& err) {
    set_ = false;
    if (value.empty()) return true;
    switch (kind_) {
        case Uint: if (!parse_uint(value, u_, err)) return false; break;
        case Float: if (!parse_float(value, f_, err)) return false; break;
        case Bool:
            if (value == "true" || value == "1") b_ = true;
            else if (value == "false" || value == "0") b_ = false;
            else { complain(err) << "boolean value expected" << std::endl; return false; }
            break;
        case Enum: if (!parse_choice(value, e_, err)) return false; break;
        case UintSet:
        case EnumSet: {
            uset_.clear();
            eset_.clear();
            size_t pos = 0;
            while (pos < value.size()) {
                size_t comma = value.find(',', pos);
                if (comma == std::string::npos) comma = value.size();
                const std::string item = first_token(value.substr(pos, comma - pos));
                if (!item.empty()) {
                    if (kind_ == UintSet) {
                        uint64_t v = 0;
                        if (!parse_uint(item, v, err)) return false;
                        uset_.insert(v);
                    } else {
                        int v = 0;
                        if (!parse_choice(item, v, err)) return false;
                        eset_.insert(v);
                    }
                }
                pos = comma + 1;
            }
            break;
        }
    }
    set_ = true;
    return true;
}

void usage(const char* program, const std::vector<Option*>& positional, const std::vector<Option*>& keyword,
           std::ostream& out) {
    out << "Usage: " << program << " [help]";
    for (const Option* o : positional) {
        if (o->required()) out << " [" << o->key() << "=]" << o->placeholder();
        else out << " [[" << o->key() << "=]" << o->placeholder() << "]";
    }
    for (const Option* o : keyword) {
        if (o->required()) out << ' ' << o->key() << '=' << o->placeholder();
        else if (o->kind() == Option::Bool && !o->as_bool()) out << " [" << o->key() << "]";
        else out << " [" << o->key() << '=' << o->placeholder() << "]";
    }
    out << std::endl;
}

bool parse(const std::vector<Option*>& positional, const std::vector<Option*>& keyword, int argc,
           const char* const* argv, std::ostream& err) {
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "help") {
            usage(argv[0], positional, keyword, err);
            return false;
        }
    auto lookup = [&](const std::string& key) -> Option* {
        for (Option* o : positional) if (o->key() == key) return o;
        for (Option* o : keyword) if (o->key() == key) return o;
        return nullptr;
    };
    size_t next_pos = 0;
    for (int i = 1; i < argc; ++i) {
        const std::string arg(argv[i]);
        const size_t eq = arg.find('=');
        if (eq != std::string::npos) {
            if (Option* o = lookup(arg.substr(0, eq)))
                if (!o->assign(first_token(arg.substr(eq + 1)), err)) return false;
            continue; // unknown keys are ignored
        }
        Option* named = lookup(arg);
        if (named && named->kind() == Option::Bool) {
            named->assign("true", err);
            continue;
        }
        while (next_pos < positional.size() && positional[next_pos]->is_set()) ++next_pos;
        if (next_pos < positional.size())
            if (!positional[next_pos]->assign(first_token(arg), err)) return false;
    }
    return true;
}

} // namespace cli
