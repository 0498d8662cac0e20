// `pathed [job.json]` — the reference's entry point (app/main.cpp:43-126) without the UI.
// Paths inside the job and the scene are relative to the current directory (the reference
// chdir("..")s out of its build directory, app/main.cpp:60, so they are repo-root relative
// there too) unless the job carries "asset_root".
#include "integrator.h"
#include "job.h"

#include <cstdio>
#include <iostream>
#include <thread>

using namespace pathed;

int runJob(const std::string &jobPath, const std::string &assetRootOverride);

int main(int argc, char *argv[])
{
    printf("Hello, world!\n");
    std::string jobPath = "job.json";
    if (argc > 1) {
        printf("Using: %s\n", argv[1]);
        jobPath = argv[1];
    }
    return runJob(jobPath, argc > 2 ? argv[2] : "");
}
