// shim for <glog/logging.h> (glog is absent in this image): LOG(x) swallows its stream.
// Lets the reference's include/logger.h -> include/match_score.h compile unmodified for oracle/_ref.
#pragma once
#include <iostream>
struct oracle_null_log { template <class T> oracle_null_log& operator<<(const T&) { return *this; } };
#define LOG(x) oracle_null_log()
