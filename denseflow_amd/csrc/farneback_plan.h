// farneback_plan.h — pure host logic of the -a=farn launches (no HIP): how the row-stream iteration kernel's column
// strips are cut into segments of rows.  Plain C++ so that the CPU tests compile the very function the launcher uses
// (tests/plan_harness.cpp, tests/test_plan_logic.py).
#pragma once

#include <algorithm>

#ifndef FARN_STREAM_GENERATIONS
#define FARN_STREAM_GENERATIONS 16
#endif

constexpr int kFarnStreamStripCols = 64; // output columns per workgroup
constexpr int kFarnStreamStepRows = 6;   // rows a workgroup advances per step
constexpr int kFarnStreamMinSegRows = 48;
constexpr int kFarnStreamSlots = 256 * 4; // workgroups the machine holds at once: 256 CUs x 4

// Rows per segment: whole 6-row steps, and enough segments that a launch is many generations of workgroups — a workgroup
// walks its whole segment, so with few generations the last, nearly empty one costs a full segment time (one segment per
// column at 1080p is 3.02 generations: measured 1130 us per launch against 1000 with >= 8).  Each segment pays 12 warm-up
// rows, hence the floor of 48 rows.  The segments [k * rows, min((k + 1) * rows, h)) partition the level's rows.
inline int farn_stream_seg_rows(int w, int h, int n_pairs) {
    const long long cols = (w + kFarnStreamStripCols - 1) / kFarnStreamStripCols;
    const long long wgs_per_seg = std::max<long long>(cols * std::max(n_pairs, 1), 1);
    long long nseg = ((long long)FARN_STREAM_GENERATIONS * kFarnStreamSlots + wgs_per_seg - 1) / wgs_per_seg;
    nseg = std::max<long long>(1, std::min<long long>(nseg, h / kFarnStreamMinSegRows)); // no segment under 48 rows
    const int rows = (int)((h + nseg - 1) / nseg);
    return (rows + kFarnStreamStepRows - 1) / kFarnStreamStepRows * kFarnStreamStepRows;
}
