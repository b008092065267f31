// colate_amd/csrc/colate_internal.h -- shared between the translation units of libcolate_amd.so
#pragma once
namespace colate {
// records the message for colate_last_error() and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
}  // namespace colate
