// ic_abi.h -- what every translation unit that defines exported (extern "C") entry points shares: the calling thread's error
// text and the exception barrier.  include/ic_amd.h promises that nothing is thrown across the ABI: every exported function
// whose body can reach an allocation, a lock or a thread is a function-try-block that ends in ICAMD_ABI_CATCH --
// std::bad_alloc and std::system_error (a thread or a lock that could not be had) become ICAMD_ERR_ALLOC, anything else
// ICAMD_ERR_HIP; the text goes to icamd_last_error().
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "ic_amd.h"

namespace icamd {
// The calling thread's last error text: a fixed buffer, so that reporting a failure (an allocation failure included) never
// allocates and never throws.
constexpr size_t kErrorChars = 256;
extern thread_local char g_last_error[kErrorChars];
void set_last_error(const char *text) noexcept;
// Stores "<what>[: <hip error name> (<hip error string>)]" and returns `code`.
int fail(int code, const char *what, hipError_t e = hipSuccess) noexcept;
// Inside a catch (...) handler: maps the exception in flight to a status (and stores its text).
int abi_exception() noexcept;
}  // namespace icamd

#define ICAMD_ABI_CATCH catch (...) { return icamd::abi_exception(); }
