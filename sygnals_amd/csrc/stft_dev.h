// Development instrumentation of stft_mel.hip (not part of the product build): with -DSYG_DEV=1 the kernel carries
// per-wave cycle accumulators per phase (s_memtime stamps) and dumps them into its statistics / mel output
// (tools/timeline.py reads them).  Such a build reports itself through syg_build_variant() and the Python binding refuses
// to load it as the product.  -DSYG_TICK_NOVM: the stamps do not drain vector memory (the DMA / store waits then show up
// where the product waits).
#pragma once
#ifndef SYG_DEV
#define SYG_DEV 0
#endif
#if SYG_DEV
#ifdef SYG_TICK_NOVM
#define SYG_TICK_WAIT "s_waitcnt lgkmcnt(0)"
#else
#define SYG_TICK_WAIT "s_waitcnt vmcnt(0) lgkmcnt(0)"
#endif
#define TICK(slot, reg)                                                                                         \
  do {                                                                                                          \
    unsigned long long _t;                                                                                      \
    asm volatile(SYG_TICK_WAIT "\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t), "+v"(reg)::"memory"); \
    tacc[slot] += _t - tprev;                                                                                   \
    tprev = _t;                                                                                                 \
  } while (0)
#define TARGS , unsigned long long (&tacc)[12], unsigned long long& tprev
#define TPASS , tacc, tprev
#else
#define TICK(slot, reg)
#define TARGS
#define TPASS
#endif
