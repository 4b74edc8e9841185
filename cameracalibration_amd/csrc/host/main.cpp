// camera_calibration_optimization <config.xml>
//
// Same command line and output files as the reference's optimiser executable
// (src/camera_calibration_optimization.cpp:560-581): one positional argument,
// results appended to ./results.txt and ./new_experiments_errors.txt.
#include "../../../include/ccal.h"

#include <cstdio>
#include <cstdlib>
#include <iostream>

int main(int argc, char** argv)
{
    if (argc < 2) {
        std::cout << "Usage: optimizer_test <config file>" << std::endl;   // :564
        std::exit(-1);
    }
    char err[1024] = {0};
    const int rc = ccal_run_config(argv[1], "results.txt", "new_experiments_errors.txt", 1, err, sizeof err);
    if (rc != CCAL_OK) {
        std::fprintf(stderr, "camera_calibration_optimization: %s\n", err[0] ? err : ccal_status_string(rc));
        return 1;
    }
    return 0;
}
