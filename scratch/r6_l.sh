#!/bin/bash
PART=pmc bash scratch/prof_round6.sh 2>&1 | tail -12
