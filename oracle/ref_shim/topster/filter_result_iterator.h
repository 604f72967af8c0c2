// ORACLE — TEST INFRASTRUCTURE ONLY. Empty stand-in for the reference's include/filter_result_iterator.h (see field.h next to it):
// topster.h needs nothing from it beyond reference_filter_result_t, which the field.h stand-in declares.
#pragma once
