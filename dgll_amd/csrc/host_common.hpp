// host_common.hpp -- what the HOST-ONLY translation units of libdgll_hip.so need (sampler.hip): the C ABI header, the export
// macro and the error plumbing.  No HIP header: these files also compile with plain g++, which is how the thread- and
// address-sanitizer builds of the threaded host code are made (tests/c_abi/Makefile; GPU sanitizers are not available).
#pragma once
#include <stdint.h>
#include <string>

#include "../../include/dgll_hip.h"

#define DGLL_API extern "C" __attribute__((visibility("default")))

namespace dgll {
// thread-local text, negative return codes; never exit()
void set_error(const std::string& msg);
}  // namespace dgll

#define DGLL_REQUIRE(cond, msg)                                     \
    do {                                                            \
        if (!(cond)) { ::dgll::set_error(std::string(msg) + " [" #cond "]"); return DGLL_ERR_INVALID; } \
    } while (0)
