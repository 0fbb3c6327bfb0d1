#!/usr/bin/env python3
"""Writes an n-row pulsar-style CSV for the reference's logistic_regression_ckks driver by cycling the committed
400-row fixture (tests/golden/pulsar_rows_head400.csv).  usage: make_lr_csv.py <rows> <out.csv>"""
import sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lines = open(os.path.join(root, "tests", "golden", "pulsar_rows_head400.csv")).read().splitlines()
head, body = lines[0], lines[1:]
n = int(sys.argv[1])
open(sys.argv[2], "w").write("\n".join([head] + [body[i % len(body)] for i in range(n)]) + "\n")
