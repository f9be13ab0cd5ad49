// tests/h5mini_harness.cpp — TEST INFRASTRUCTURE: C wrappers around include/h5mini.h for ctypes.
#include <cstring>
#include <string>

#include "../include/h5mini.h"

extern "C" {
int h5h_create(const char *path, char *err, int cap) {
    try {
        h5mini::create(path);
        return 0;
    } catch (const std::exception &e) {
        strncpy(err, e.what(), cap - 1);
        return -1;
    }
}
// n datasets named names[i], all rows x cols, data = n*rows*pitch_floats floats
int h5h_append(const char *path, int n, const char *const *names, int rows, int cols, int pitch_floats,
               const float *data, char *err, int cap) {
    try {
        std::vector<h5mini::FloatDataset> ds;
        for (int i = 0; i < n; ++i)
            ds.push_back({names[i], (size_t)rows, (size_t)cols, data + (size_t)i * rows * pitch_floats,
                          (size_t)pitch_floats * sizeof(float)});
        h5mini::append(path, ds);
        return 0;
    } catch (const std::exception &e) {
        strncpy(err, e.what(), cap - 1);
        return -1;
    }
}
int h5h_count(const char *path) {
    try {
        return (int)h5mini::list(path).size();
    } catch (...) {
        return -1;
    }
}
}
