// Shared by run_core.cpp (the dgr_* entry points) and the backend translation unit linked beside it.
#pragma once
#include <string>
#include <vector>

#include "../../include/dipgenie_run.h"
#include "pipeline.hpp"

struct dgr_handle {
    dg::Pipeline p;
    std::string hap_buf;
    int32_t hap_buf_h = -1;
    std::string read_bases;
    std::vector<int64_t> read_off;
    void *ctx = nullptr;              // backend context (product: dg_ctx)
    int device = 0;
};

// Fills H->p.be (creating the backend context on first use); != 0 with a message when no backend is usable -- there is no fallback.
int dgr_wire_backend(dgr_handle *H, std::string &err);
void dgr_unwire_backend(dgr_handle *H);
