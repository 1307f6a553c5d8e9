#!/bin/bash
echo "1 process x 16 engines"; python tools/concurrent_climbs.py --engines 16 2>&1 | tail -1 | cut -c1-100
echo "2 processes x 8 engines"; for i in 1 2; do python tools/concurrent_climbs.py --engines 8 2>&1 | tail -1 | cut -c1-100 & done; wait
echo "4 processes x 4 engines"; for i in 1 2 3 4; do python tools/concurrent_climbs.py --engines 4 2>&1 | tail -1 | cut -c1-100 & done; wait
echo "8 processes x 2 engines"; for i in 1 2 3 4 5 6 7 8; do python tools/concurrent_climbs.py --engines 2 --climbs 2 2>&1 | tail -1 | cut -c1-100 & done; wait
