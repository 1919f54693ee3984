"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference hot path + golden generation.
Nothing under dinounet_amd/ imports this package."""
