#!/bin/bash
# Both measurement passes of a round, one after the other (run through gpurun): see profile_round.sh / profile_round2.sh.
bash scripts/profile_round.sh && bash scripts/profile_round2.sh
