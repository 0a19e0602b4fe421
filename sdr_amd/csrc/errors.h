// errors.h -- per-thread last-error text behind sdr_last_error().
#pragma once
#include <string>

namespace sdr_amd {
// Records `msg` for sdr_last_error() and returns `code`.
int fail(int code, const std::string& msg);
void clear_error();
} // namespace sdr_amd
