// include/h5mini.h — a dependency-free writer for the small subset of HDF5 the reference's -st=h5 output uses.
//
// The reference (built with USE_HDF5=ON) creates `<output_dir>.h5` per video with H5Fcreate
// (/root/reference/src/denseflow_gpu.cpp:223-243) and, per FlowBuffer, re-opens it and adds one float dataset
// per flow component with H5LTmake_dataset_float: `/flow_x_%05d`, `/flow_y_%05d` (`_p%d_` / `_m%d_` infixes for
// other steps), rank 2, rows x cols (src/common.cpp:121-149, src/utils.cpp:28-41).  libhdf5 is not part of this
// environment's toolchain contract, so the files are written here directly in the classic on-disk
// format libhdf5 1.8/1.10 itself produces for such files: version-0 superblock, root group as symbol table
// (v1 B-tree + local heap + SNOD nodes), version-1 object headers with dataspace / datatype (IEEE f32 LE) /
// fill-value / contiguous-layout messages.  Files are readable by any HDF5 library (tests read them back with
// libhdf5 where it exists and with an independent parser everywhere).
#pragma once

#include <cstddef>
#include <string>
#include <vector>

namespace h5mini {

struct FloatDataset {
    std::string name;  // link name in the root group, without the leading '/'
    size_t rows, cols; // rank-2 dataspace {rows, cols}
    const float *data; // rows x cols floats, row pitch `pitch_bytes`
    size_t pitch_bytes;
};

// H5Fcreate(path, H5F_ACC_TRUNC) + H5Fclose: an empty file with a root group.
void create(const std::string &path);

// H5Fopen(RDWR) + H5LTmake_dataset_float per entry + H5Fclose.  Throws std::runtime_error on I/O errors, on a
// name that already exists (as H5LTmake_dataset_float fails) and on files outside the subset written here.
void append(const std::string &path, const std::vector<FloatDataset> &datasets);

// Names of the root group's links in on-disk (sorted) order — used by tests and by append itself.
std::vector<std::string> list(const std::string &path);

} // namespace h5mini
