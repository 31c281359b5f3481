#!/bin/bash
# final profile refresh of the round: bench line, kernel trace summary, PMC traffic (bench workload, eager token loop)
set -o pipefail
bash profiles/collect.sh r02 2>&1 | tail -5
