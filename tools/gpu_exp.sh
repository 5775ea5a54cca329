#!/bin/bash
timeout 2700 python -m pytest tests -m gpu -q -x 2>&1 | tail -4 | cut -c1-250
