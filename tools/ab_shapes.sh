#!/bin/bash
# tools/ab_configs.sh restricted to shapes matching a pattern:  tools/ab_shapes.sh 'PATTERN' NAME [NAME ...]
pat=$1; shift
bash "$(dirname "$0")/ab_configs.sh" "$@" | grep -E "$pat"
