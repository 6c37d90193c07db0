"""TEST INFRASTRUCTURE ONLY (see oracle/crc_oracle.h): restatement of the control flow of the reference's plain-modulus
binary search, used to pin crcnn_amd/host/plain_modulus_search.cpp.  Follows CrCNN/src/optimalParametersChooser.cpp
statement by statement (line numbers below), including its conversions through log2() truncated to an integer.
Pinned itself: tests/test_search_logic.py demands that `internal` reproduce every candidate sequence recorded from the REFERENCE's
compiled plainModulusBinarySearchInternal (tests/golden/search_sequences.json <- oracle/_ref/search_harness).  Its remaining use is to replay
the recursion over verdicts observed on the GPU (tests/test_gpu_host_cpp.py), which no pre-recorded table can anticipate."""
import math

SUCCESS, OUT_OF_BUDGET, MISPREDICTED = "SUCCESS", "OUT_OF_BUDGET", "MISPREDICTED"


def internal(test, lo, hi, pow_):                       # plainModulusBinarySearchInternal :73-180
    assert lo <= hi                                     # :75
    if pow_:                                            # :87-90
        lo = int(math.log2(lo)); hi = int(math.log2(hi))
    if hi - lo <= 1:                                    # :91
        if pow_:                                        # :92-95
            lo = 1 << lo; hi = 1 << hi
        s = test(lo)                                    # :100
        if s == SUCCESS:                                # :102-106
            return lo
        if s == OUT_OF_BUDGET:                          # :107-111
            return 0
        if pow_:                                        # :112-115
            lo = int(math.log2(lo)); hi = int(math.log2(hi))
        if hi - lo == 1:                                # :116
            if pow_:
                hi = 1 << hi
            return hi if test(hi) == SUCCESS else 0     # :120-126
        return 0                                        # :128
    t = lo + (hi - lo) // 2                             # :131
    if pow_:                                            # :132-136
        t = 1 << t; lo = 1 << lo; hi = 1 << hi
    s = test(t)                                         # :142
    if s in (SUCCESS, OUT_OF_BUDGET):                   # :149
        r = internal(test, lo, t - 1, pow_)             # :150
        if r > 0:                                       # :152-156
            return r
        return t if s == SUCCESS else 0                 # :157-164
    if t >= hi:                                         # :168-172
        return 0
    return internal(test, t + 1, hi, pow_)              # :174


def search(test, lo, hi, min_q):                        # plainModulusBinarySearch :30-60
    found = internal(test, lo, hi, True)                # :46
    if found > 0 and found >= min_q:                    # :52
        hi2 = min_q - 1                                 # :57
        lo2 = 1 << int(math.floor(math.log2(min_q)))    # :58
        fast = internal(test, lo2, hi2, False)          # :61
        if fast > 0:                                    # :62-63
            return fast
    return found                                        # :65
