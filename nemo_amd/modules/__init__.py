from .audio_preprocessing import AudioToMelSpectrogramPreprocessor, FilterbankFeatures  # noqa: F401
