// timer.h -- the two timer shapes the simulation API exposes (ref: timer.h:6-91):
// IntervallTimer (elapsed time on request) and the *ScopeTimer family (print on scope
// exit).  One steady-clock template underneath.
#pragma once
#include <chrono>
#include <cstdint>
#include <iostream>

namespace nbody_host_detail {
template <typename Unit>
inline int64_t Elapsed(std::chrono::steady_clock::time_point since) {
  return std::chrono::duration_cast<Unit>(std::chrono::steady_clock::now() - since).count();
}
template <typename Unit>
class PrintingScopeTimer {
 public:
  explicit PrintingScopeTimer(const char* unit_name) : unit_name_(unit_name) {}
  ~PrintingScopeTimer() { std::cout << Elapsed<Unit>(start_) << " " << unit_name_ << std::endl; }
 private:
  std::chrono::steady_clock::time_point start_ = std::chrono::steady_clock::now();
  const char* unit_name_;
};
}  // namespace nbody_host_detail

class IntervallTimer {
 public:
  int64_t getTimeInMinutes() { return nbody_host_detail::Elapsed<std::chrono::minutes>(start_); }
  int64_t getTimeInSeconds() { return nbody_host_detail::Elapsed<std::chrono::seconds>(start_); }
  int64_t getTimeInMilliseconds() { return nbody_host_detail::Elapsed<std::chrono::milliseconds>(start_); }
  int64_t getTimeInMicroseconds() { return nbody_host_detail::Elapsed<std::chrono::microseconds>(start_); }
  int64_t getTimeInNanoseconds() { return nbody_host_detail::Elapsed<std::chrono::nanoseconds>(start_); }
 private:
  std::chrono::steady_clock::time_point start_ = std::chrono::steady_clock::now();
};

struct MinutesScopeTimer : nbody_host_detail::PrintingScopeTimer<std::chrono::minutes> {
  MinutesScopeTimer() : PrintingScopeTimer("Minutes") {}
};
struct SecondsScopeTimer : nbody_host_detail::PrintingScopeTimer<std::chrono::seconds> {
  SecondsScopeTimer() : PrintingScopeTimer("Seconds") {}
};
struct MillisecondsScopeTimer : nbody_host_detail::PrintingScopeTimer<std::chrono::milliseconds> {
  MillisecondsScopeTimer() : PrintingScopeTimer("Milliseconds") {}
};
struct MicrosecondsScopeTimer : nbody_host_detail::PrintingScopeTimer<std::chrono::microseconds> {
  MicrosecondsScopeTimer() : PrintingScopeTimer("Microseconds") {}
};
struct NanosecondsScopeTimer : nbody_host_detail::PrintingScopeTimer<std::chrono::nanoseconds> {
  NanosecondsScopeTimer() : PrintingScopeTimer("Nanoseconds") {}
};
