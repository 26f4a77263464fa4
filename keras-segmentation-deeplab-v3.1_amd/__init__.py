"""MI355X-native DeepLabV3+ forward/backward path."""
