// The one switch between the product sources and the experiments build (make EXP=1 -> -DCF_EXPERIMENTS -> libcenterface_hip_exp.so).
// Measured-and-rejected kernel variants, A/B table rows and the two-lane schedule live under experiments/ (*.inc fragments cut out of
// the file named in their first line, experiments/*.hip whole translation units); a product source only carries the hook
//     #include CF_EXP_INC(name)
// which includes experiments/name.inc in an experiments build and an empty file in the release build: what the release sources
// show is exactly what ships.  CF_EXP_ON is 1 / 0 for the few places where the release build has code the experiments build replaces.
#pragma once
#define CF_EXP_STR_(x) #x
#define CF_EXP_STR(x) CF_EXP_STR_(x)
#ifdef CF_EXPERIMENTS
#define CF_EXP_ON 1
#define CF_EXP_INC(name) CF_EXP_STR(experiments/name.inc)
#else
#define CF_EXP_ON 0
#define CF_EXP_INC(name) "cf_exp_none.inc"
#endif
