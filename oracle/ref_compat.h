// Force-included (-include) when building the UNMODIFIED reference sources with g++ (oracle/_ref).
// TEST INFRASTRUCTURE ONLY - never linked into the product library.
// The reference is MSVC code; this header supplies the headers MSVC pulls in transitively and makes the
// unqualified math calls resolve to the float overloads, as they do on MSVC.
// No reference text is copied here and no reference file is patched or duplicated.
#pragma once
#ifdef __cplusplus
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <stdexcept>
#include <string>
using std::sqrt; using std::pow; using std::abs; using std::isnan;
using std::sin; using std::cos; using std::tan; using std::atan2; using std::asin;
#ifdef REF_COMPAT_MSVC_EXCEPTION_CTOR
// MSVC's std::exception has a (const char*) constructor; libstdc++'s does not. Pull in every header the
// translation unit uses first (their include guards then skip them), and only then map the two
// `throw std::exception("...")` sites onto std::runtime_error.
#include <iostream>
#include <fstream>
#include <vector>
#include <array>
#include <glm/glm.hpp>
#define exception runtime_error
#endif
#endif
