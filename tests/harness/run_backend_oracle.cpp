// TEST INFRASTRUCTURE: the checker (oracle/liboracle.so) behind the Backend table of the run library, so that the host pipeline
// and the sharded run can be tested on a machine without a GPU (libdg_run_oracle.so).  Never shipped, never linked by anything
// under dipgenie_amd/.
#include <cstdlib>

#include "../../dipgenie_amd/host/run_core.hpp"
#include "../../oracle/oracle.h"

static int o_sketch_reads(void *, const char *b, const int64_t *off, int64_t n, int k, int w, uint64_t **h, int32_t **c, int64_t *nd) { return orc_sketch_reads(b, off, n, k, w, h, c, nd); }
static int o_sketch_hap(void *, const char *s, int64_t len, int k, int w, uint64_t **h, int64_t **p, int64_t *n) {
    const int64_t cnt = orc_minimizers(s, len, k, w, nullptr, nullptr, 0);
    *h = (uint64_t *)malloc(sizeof(uint64_t) * (cnt + 1));
    *p = (int64_t *)malloc(sizeof(int64_t) * (cnt + 1));
    *n = orc_minimizers(s, len, k, w, *h, *p, cnt);
    return 0;
}
static int o_dp(void *, const dg_dp_graph *g, dg_dp_result *r) { return orc_dp_solve_diploid((const orc_dp_graph *)g, (orc_dp_result *)r, nullptr); }
static const char *o_err() { return "oracle"; }

int dgr_wire_backend(dgr_handle *H, std::string &) {
    dg::Pipeline &p = H->p;
    p.be.sketch_reads = o_sketch_reads; p.be.sketch_haplotype = o_sketch_hap; p.be.dp_solve_diploid = o_dp; p.be.free_buf = orc_free; p.be.last_error = o_err;
    return 0;
}

void dgr_unwire_backend(dgr_handle *) {}
