"""pair-net_amd: MI355X-native Pair-Net inference hot path (import as `pairnet_amd`)."""
