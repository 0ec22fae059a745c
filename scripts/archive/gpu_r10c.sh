#!/bin/bash
# r10c: the long-list pass skipped, primitive by primitive, what lies behind the hits a tile already holds (the depth class in the spare bits of the rectangle's
# y1, one v_readlane + a compare per candidate before its slab test) against the round-level stop only (nohidden): 7 % of the slab tests went away, the pass got
# no shorter -- the candidates a tile meets are mostly not hidden ones (horizon tiles with a sky pixel never stop).  Not kept; the code is in the history only
# (this script's variant build no longer exists).
echo "r10c: see profiles/r10_experiments.txt"
