python3 tools/pcie_inclusive.py 2>&1 | tail -3
python3 tools/partition_probe.py 2>&1 | tail -6
