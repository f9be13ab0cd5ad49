// src/utils.cpp — see include/utils.h
#include "utils.h"

void createFile(const path &ph) { std::ofstream(ph.string()).close(); }

double CurrentSeconds() {
    using namespace std::chrono;
    return duration_cast<milliseconds>(system_clock::now().time_since_epoch()).count() / 1000.0;
}
