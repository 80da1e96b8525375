"""tracknetv3_amd -- MI355X-native hot path of TrackNetV3."""
