"""krotov_amd -- MI355X-native Krotov optimal-control engine."""
