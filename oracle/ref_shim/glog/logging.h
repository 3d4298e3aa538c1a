// ORACLE — TEST INFRASTRUCTURE ONLY.  glog stand-in: LOG(x) << ... is swallowed, LOG(FATAL) aborts.  While ref_shim::log_capture points
// at a string, everything streamed into LOG(...) is appended to it instead: FullSystem::optimize reports its per-iteration energies only
// through printOptRes -> LOG(INFO) (FullSystem.cc:1795-1807), and the pin reads them from there (ref_driver.cc: ref_fs_optimize).
#pragma once
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <utility>
namespace ref_shim {
// (per thread: the tracking thread and the mapping thread of tests/test_adapter_threads_gpu.py log side by side; the reference's IndexThreadReduce workers have no slot and stay silent)
inline std::string *&log_capture_slot() { static thread_local std::string *p = nullptr; return p; }
template <class T, class = void> struct is_streamable : std::false_type {};
template <class T> struct is_streamable<T, std::void_t<decltype(std::declval<std::ostream &>() << std::declval<const T &>())>> : std::true_type {};
struct NullLog {
    bool fatal; std::ostringstream *os;
    explicit NullLog(bool f) : fatal(f), os(log_capture_slot() ? new std::ostringstream() : nullptr) {}
    ~NullLog() {
        if (os) { if (log_capture_slot()) *log_capture_slot() += os->str(); delete os; }
        if (fatal) { std::cerr << "LOG(FATAL) in reference code" << std::endl; std::abort(); }
    }
    template <class T> NullLog &operator<<(const T &v) { if constexpr (is_streamable<T>::value) { if (os) *os << v; } return *this; }
    NullLog &operator<<(std::ostream &(*f)(std::ostream &)) { if (os) *os << f; return *this; }
};
}
#define INFO 0
#define WARNING 1
#define ERROR 2
#define FATAL 3
#define LOG(sev) ref_shim::NullLog((sev) == FATAL)
#define VLOG(n) ref_shim::NullLog(false)
#define DLOG(sev) ref_shim::NullLog(false)
#define CHECK(c) if (!(c)) ref_shim::NullLog(true)
