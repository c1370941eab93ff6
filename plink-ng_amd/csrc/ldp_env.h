// ldp_env.h -- the one place libldprune_hip.so may look at its environment.
//
// The shipped library never does: LDP_ENV(name) is a null pointer there, the string does not even reach the binary
// (tests/test_cabi_symbols.py greps for it), so no variable can change which kernel runs or what it computes.  Tuning aids,
// timing prints and the ablation instantiations of the pair kernels (which compute WRONG results by construction) exist only in
// the measurement build, `-DLDP_MEASURE` -> lib/libldprune_hip_measure.so (plink_ng_amd.build_library(measure=True), used by tools/).
// What the parity tests need to steer -- many small decode launches, the replay in instalments, ... -- are per-engine options of
// ldp_debug_set_option() (include/ldprune_hip_debug.h), not process state.
#ifndef LDP_ENV_H
#define LDP_ENV_H
#ifdef LDP_MEASURE
#include <cstdlib>
#define LDP_ENV(name) getenv(name)
#else
#define LDP_ENV(name) (static_cast<const char*>(nullptr))
#endif
#endif
