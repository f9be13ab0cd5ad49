// include/utils.h — small helpers with the reference's names (/root/reference/include/utils.h).
#ifndef DENSEFLOW_UTILS_H
#define DENSEFLOW_UTILS_H

#include <sys/stat.h>

#include "common.h"

double CurrentSeconds();          // wall clock, millisecond resolution
void createFile(const path &ph);  // touch

inline bool fileExists(const string &name) {
    struct stat info;
    return stat(name.c_str(), &info) == 0;
}
inline bool dirExists(const string &p) {
    struct stat info;
    return stat(p.c_str(), &info) == 0 && (info.st_mode & S_IFDIR);
}
#endif // DENSEFLOW_UTILS_H
