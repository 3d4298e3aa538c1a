// ORACLE — TEST INFRASTRUCTURE ONLY.  glog stand-in: LOG(x) << ... is swallowed, LOG(FATAL) aborts.
#pragma once
#include <cstdlib>
#include <iostream>
namespace ref_shim {
struct NullLog { bool fatal; explicit NullLog(bool f) : fatal(f) {} ~NullLog() { if (fatal) { std::cerr << "LOG(FATAL) in reference code" << std::endl; std::abort(); } }
    template <class T> NullLog &operator<<(const T &) { return *this; }
    NullLog &operator<<(std::ostream &(*)(std::ostream &)) { return *this; } };
}
#define INFO 0
#define WARNING 1
#define ERROR 2
#define FATAL 3
#define LOG(sev) ref_shim::NullLog((sev) == FATAL)
#define VLOG(n) ref_shim::NullLog(false)
#define DLOG(sev) ref_shim::NullLog(false)
#define CHECK(c) if (!(c)) ref_shim::NullLog(true)
