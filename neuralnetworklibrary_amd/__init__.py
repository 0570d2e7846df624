"""MI355X-native Learner.fit() hot path behind the NeuralNetworkLibrary API."""
__version__ = '0.1.0'
