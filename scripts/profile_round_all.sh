#!/bin/bash
# The measurement passes of a round, one after the other (run through gpurun): see profile_round.sh / profile_round2.sh /
# profile_round3.sh.  The PMC pass (round2) writes profiles/traffic.json on the box; round3's bench line then carries it.
bash scripts/profile_round.sh && bash scripts/profile_round2.sh && bash scripts/profile_round3.sh
