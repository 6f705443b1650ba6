"""Test package: CPU suite (oracle vs goldens, C-ABI, kernel sources through the wave simulator, harness, gloo DP) and the
``gpu``-marked parity suite that calls libtic_hip.so on an MI355X."""
