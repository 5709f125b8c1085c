// Translation unit of the cf_mbconv3.hip kernel instances that are faster under the ILP-first machine scheduler
// (Makefile: EXTRA_cf_mbconv3_ilp = -mllvm -amdgpu-sched-strategy=max-ilp; the list and the measurements are in cf_mbconv3.hip).
#define CF_ILP_TU 1
#include "cf_mbconv3.hip"
